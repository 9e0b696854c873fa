// dfft_generic.hip -- run-time-scheduled Stockham kernel for every 7-smooth length up to 4096 that has no tuned plan in
// dfft_plans.h.
//
// The reference JIT-compiles a kernel for any product of radices 2/3/4/5/7/8 (templateFFT.cpp:3941-4607 FFTScheduler,
// single pass up to 4096 points).  The tuned static table covers the lengths on the benchmarked paths; this kernel gives
// the remaining lengths (20, 40, 80, 160, 200, 320, 640, 1000, 1280, 1536, 3072, 4096, ...) the same functionality --
// same address maps (pack fused into the Y pass, transposed store of the X pass, uneven slabs, plane-chunked launches,
// folded scaling), so the slab pipeline needs no special case -- at a lower, un-tuned rate:
//   * the whole tile (N points x CB columns) lives in LDS, two buffers, one Stockham stage = one LDS round trip;
//   * the radix sequence is a kernel argument, the butterflies are the same Butterfly<R> templates the tuned kernels use;
//   * loads/stores walk the tile in the order that is contiguous in memory on that side (columns fastest, or the FFT
//     index fastest for the transposed side), so both sides stay coalesced.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "dfft_butterfly.h"
#include "dfft_kernels.h"

namespace dfft {

struct RadixSchedule {
    int n;  // stages
    int r[12];
};

static bool make_schedule(int N, RadixSchedule& s) {
    s.n = 0;
    if (N < 2) return false;
    int m = N;
    for (int f : {8, 4, 2, 7, 5, 3}) {
        while (m % f == 0) {
            if (s.n >= 12) return false;
            s.r[s.n++] = f;
            m /= f;
        }
    }
    return m == 1;
}

bool generic_length_supported(int n) {
    RadixSchedule s;
    return n >= 2 && n <= 4096 && make_schedule(n, s);
}

namespace {

__device__ __forceinline__ long long map_offset(const AxisMap& m, int idx, int c, long long a) {
    const int ib = m.nblk == 1 ? 0 : idx / m.blk;
    long long off = (m.sub > 1 ? (long long)(ib / m.sub) * m.blk_stride + (long long)(ib % m.sub) * m.sub_stride
                               : (long long)ib * m.blk_stride) +
                    (long long)(idx - ib * m.blk) * m.stride + (long long)c * m.cstride;
    if (ib == m.nblk - 1) off += a * m.last_delta;
    return off;
}

template <int R, int DIR, class V>
__device__ __forceinline__ void stage_butterfly(const V* src, V* dst, const V* __restrict__ tw, int N, int CB, int Ns, int ns_shift,
                                                int step, int j, int c) {
    V         u[R];
    // Ns is a power of two for all the radix-8/4/2 stages (they come first in the schedule): shift/mask, no division
    const int jq = ns_shift >= 0 ? (j >> ns_shift) : j / Ns;
    const int m = j - jq * Ns;
#pragma unroll
    for (int r = 0; r < R; ++r) u[r] = src[(j + r * (N / R)) * CB + c];
    if (Ns > 1) {
#pragma unroll
        for (int r = 1; r < R; ++r) {
            V w = tw[(r * m) * step];
            if (DIR < 0) w.y = -w.y;
            u[r] = cmul(u[r], w);
        }
    }
    Butterfly<R, DIR, V>::run(u);
    const int base = jq * Ns * R + m;
#pragma unroll
    for (int r = 0; r < R; ++r) dst[(base + r * Ns) * CB + c] = u[r];
}

template <class V, int DIR>
__global__ void __launch_bounds__(1024) fft_generic_kernel(const V* in, V* out, const V* __restrict__ tw, AxisMap imap, AxisMap omap,
                                                          TileMap itile, TileMap otile, unsigned ntiles, unsigned tiles_per_a,
                                                          int ncols, unsigned a_first, double scale, int N, int CB,
                                                          RadixSchedule sched) {
    extern __shared__ __attribute__((aligned(16))) char dfft_gsmem[];
    V*        buf0 = reinterpret_cast<V*>(dfft_gsmem);
    V*        buf1 = buf0 + (size_t)N * CB;
    const int tid = threadIdx.x, nthr = blockDim.x, tile_elems = N * CB, cb_shift = __builtin_ctz(CB);
    const bool in_idx_fast = imap.stride == 1 && imap.cstride != 1, out_idx_fast = omap.stride == 1 && omap.cstride != 1;
    using Rt = typename real_of<V>::type;
    for (unsigned t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const unsigned al = t / tiles_per_a, b = t - al * tiles_per_a;
        const long long a = (long long)al + a_first;
        const V* ip = in + a * itile.a_stride + (long long)b * CB * itile.b_stride;
        V*       op = out + a * otile.a_stride + (long long)b * CB * otile.b_stride;
        for (int e = tid; e < tile_elems; e += nthr) {
            const int idx = in_idx_fast ? e % N : e >> cb_shift, c = in_idx_fast ? e / N : e & (CB - 1);
            V         v = V{0, 0};
            if ((int)(b * CB) + c < ncols) v = ip[map_offset(imap, idx, c, a)];
            buf0[idx * CB + c] = v;
        }
        __syncthreads();
        V*  src = buf0;
        V*  dst = buf1;
        int Ns = 1;
        for (int s = 0; s < sched.n; ++s) {
            const int R = sched.r[s], work = (N / R) * CB, step = N / (Ns * R);
            const int sh = (Ns & (Ns - 1)) == 0 ? __builtin_ctz(Ns) : -1;
            for (int w = tid; w < work; w += nthr) {
                const int j = w >> cb_shift, c = w & (CB - 1);  // CB is a power of two
                switch (R) {
                    case 2: stage_butterfly<2, DIR, V>(src, dst, tw, N, CB, Ns, sh, step, j, c); break;
                    case 3: stage_butterfly<3, DIR, V>(src, dst, tw, N, CB, Ns, sh, step, j, c); break;
                    case 4: stage_butterfly<4, DIR, V>(src, dst, tw, N, CB, Ns, sh, step, j, c); break;
                    case 5: stage_butterfly<5, DIR, V>(src, dst, tw, N, CB, Ns, sh, step, j, c); break;
                    case 7: stage_butterfly<7, DIR, V>(src, dst, tw, N, CB, Ns, sh, step, j, c); break;
                    default: stage_butterfly<8, DIR, V>(src, dst, tw, N, CB, Ns, sh, step, j, c); break;
                }
            }
            __syncthreads();
            Ns *= R;
            V* tmp = src;
            src = dst;
            dst = tmp;
        }
        const Rt sc = (Rt)scale;
        for (int e = tid; e < tile_elems; e += nthr) {
            const int idx = out_idx_fast ? e % N : e >> cb_shift, c = out_idx_fast ? e / N : e & (CB - 1);
            if ((int)(b * CB) + c < ncols) {
                V v = src[idx * CB + c];
                if (scale != 1.0) v = cscale(v, sc);
                op[map_offset(omap, idx, c, a)] = v;
            }
        }
        __syncthreads();  // the next tile's loads overwrite buf0
    }
}

template <class V> hipError_t launch_generic_t(const FftLaunch& Lin, hipStream_t stream) {
    RadixSchedule sched;
    if (!make_schedule(Lin.n, sched)) return hipErrorInvalidValue;
    FftLaunch L = Lin;
    const int N = L.n;
    if (!L.cols) {
        // rows = columns of the transposed view: FFT index unit-stride, "column" c = row within the tile
        L.imap = L.omap = AxisMap{N, 1, 0, 1, (long long)N, 0, 1, 0};
        L.itile = L.otile = TileMap{0, (long long)N};
        L.in = (const char*)Lin.in + (size_t)Lin.a_first * N * sizeof(V);
        L.out = (char*)Lin.out + (size_t)Lin.a_first * N * sizeof(V);
        L.a_first = 0;
        if (Lin.ntiles >= (1ll << 31)) return hipErrorInvalidValue;
        L.ncols = (int)Lin.ntiles;
        L.na = 1;
    }
    // columns per tile: full 128-byte lines, more for short FFTs (a tile should hold a few thousand points), never more
    // than two tile buffers of LDS allow
    const int line = 128 / (int)sizeof(V);
    int       cb = L.cols ? line : 1;
    while (cb * 2 <= 64 && N * cb < 2048) cb *= 2;
    while (cb > 1 && (size_t)2 * N * cb * sizeof(V) > 128 * 1024) cb /= 2;
    const size_t lds = (size_t)2 * N * cb * sizeof(V);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    L.tiles_per_a = (L.ncols + cb - 1) / cb;
    L.ntiles = L.na * L.tiles_per_a;
    if (L.ntiles <= 0) return hipSuccess;
    if (L.ntiles >= (1ll << 31)) return hipErrorInvalidValue;
    auto kern = L.dir > 0 ? fft_generic_kernel<V, +1> : fft_generic_kernel<V, -1>;
    static std::atomic<bool> attr_set[2][64];
    static std::mutex        setup_mutex;
    int         dev = 0;
    hipError_t  e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[L.dir > 0][dev].load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lk(setup_mutex);
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set[L.dir > 0][dev].store(true, std::memory_order_release);
    }
    static thread_local int cached_dev = -1, cus = 256;
    if (cached_dev != dev) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        cached_dev = dev;
    }
    int bpc = (int)((size_t)160 * 1024 / lds);
    if (bpc > 8) bpc = 8;
    if (bpc < 1) bpc = 1;
    long long grid = (long long)cus * bpc;
    if (grid > L.ntiles) grid = L.ntiles;
    (void)hipGetLastError();
    // about four tile elements per thread (measured: 256 threads best for 1280-element tiles, 512 for 2560, 1024 from 4000)
    int threads = ((N * cb / 4 + 63) / 64) * 64;
    threads = threads < 256 ? 256 : (threads > 1024 ? 1024 : threads);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(threads), lds, stream, (const V*)L.in, (V*)L.out, (const V*)L.tw, L.imap,
                       L.omap, L.itile, L.otile, (unsigned)L.ntiles, (unsigned)L.tiles_per_a, L.ncols, (unsigned)L.a_first,
                       L.scale == 0.0 ? 1.0 : L.scale, N, cb, sched);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_generic(const FftLaunch& L, hipStream_t stream) {
    if (L.rot.in_mode != 0 || L.rot.out_mode != 0) return hipErrorInvalidValue;  // rotated rows: tuned kernels only
    if (L.dtype == F64) return launch_generic_t<double2>(L, stream);
    if (L.dtype == F32) return launch_generic_t<float2>(L, stream);
    return hipErrorInvalidValue;
}

}  // namespace dfft
