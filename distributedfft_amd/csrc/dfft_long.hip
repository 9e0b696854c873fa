// dfft_long.hip -- axes longer than the single-pass range (> 4096 points): the four-step decomposition on top of the
// single-pass kernels.
//
// Replaces (behaviour) the reference generator's multi-upload plans: templateFFT.cpp:3972-4106 (FFTScheduler splitting an
// axis into several "axis uploads") and the reorder pass behind its LUT (:5144-5153).  Not on the benchmarked path -- no
// BASELINE configuration has an axis above 4096 -- so this is the plain textbook form, correct for every length
// N = N1 * N2 whose factors have tuned single-pass plans, not a tuned one (4 passes over the data instead of 1):
//   data[b][n][s]  (b batch, n the FFT axis, s contiguous columns; contiguous rows are s = 1),  n = n1 * N2 + n2
//     A  FFT over n1 (N1 points, stride N2*s) for every (n2, c)                      in      -> scratch   (column kernel)
//     T  scratch[b][k1][n2][c] *= W_N^{k1 * n2}                                       scratch              (this file)
//     B  FFT over n2 (N2 points, stride s) for every (k1, c)                          scratch              (column / row kernel)
//     P  out[b][k2][k1][c] = scale * scratch[b][k1][k2][c]   (X[k1 + N1 k2])          scratch -> out       (this file)
// W_N^m is looked up in two small tables, W_N^m = hi[m / 4096] * lo[m % 4096] (one complex multiplication), so no table
// grows with N.
#include <mutex>
#include <map>
#include <memory>
#include <cmath>
#include <vector>

#include <algorithm>
#include <cstring>
#include <tuple>

#include "dfft_butterfly.h"
#include "dfft_internal.h"
#include "dfft_long.h"

namespace dfft {

namespace {

constexpr int kLoBits = 12, kLo = 1 << kLoBits;

template <class V> __device__ __forceinline__ V cmulv(V a, V b) { return V{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

// T: e = ((b * N1 + k1) * N2 + n2) * s + c
template <class V>
__global__ void __launch_bounds__(256) long_twiddle_kernel(V* data, long long total, int N1, int N2, long long s, const V* __restrict__ hi,
                                                           const V* __restrict__ lo, int dir) {
    const long long rowlen = (long long)N2 * s;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / rowlen;           // b * N1 + k1
        const int       k1 = (int)(r % N1);
        const int       n2 = (int)((e - r * rowlen) / s);
        const long long m = (long long)k1 * n2;   // < N1 * N2
        V w = cmulv(hi[m >> kLoBits], lo[m & (kLo - 1)]);
        if (dir < 0) w.y = -w.y;
        data[e] = cmulv(data[e], w);
    }
}

// P, s > 1: one (k1, k2) pair = one run of s contiguous elements; a wave copies runs, lanes along c
template <class V>
__global__ void __launch_bounds__(256) long_permute_runs_kernel(const V* in, V* out, long long nruns, int N1, int N2, long long s,
                                                                typename real_of<V>::type scale) {
    const int       lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
    const long long per = (long long)N1 * N2;
    for (long long r = wave; r < nruns; r += nw) {  // r = (b * N1 + k1) * N2 + k2  (source order)
        const long long b = r / per, q = r - b * per;
        const int       k1 = (int)(q / N2), k2 = (int)(q - (long long)k1 * N2);
        const V*        src = in + r * s;
        V*              dst = out + (b * per + (long long)k2 * N1 + k1) * s;
        for (long long c = lane; c < s; c += 64) {
            V v = src[c];
            v.x *= scale;
            v.y *= scale;
            dst[c] = v;
        }
    }
}

// P, s == 1: batched 32 x 32 tile transpose of [N1][N2] matrices through padded LDS
template <class V>
__global__ void __launch_bounds__(256) long_permute_tile_kernel(const V* in, V* out, int N1, int N2, int tiles_c,
                                                                typename real_of<V>::type scale) {
    __shared__ V    tile[32][33];
    const int       lane = threadIdx.x & 31, w = threadIdx.x >> 5;
    const int       tr = blockIdx.x / tiles_c, tc = blockIdx.x - tr * tiles_c;
    const long long base = (long long)blockIdx.y * N1 * N2;
    const int       r0 = tr * 32, c0 = tc * 32;
    for (int i = w; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + lane;
        if (r < N1 && c < N2) tile[i][lane] = in[base + (long long)r * N2 + c];
    }
    __syncthreads();
    for (int i = w; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + lane;
        if (r < N1 && c < N2) {
            V v = tile[lane][i];
            v.x *= scale;
            v.y *= scale;
            out[base + (long long)c * N1 + r] = v;
        }
    }
}

struct LongTw {
    void *hi, *lo;
};
std::mutex                                        g_mutex;
std::map<std::tuple<int, long long, int>, LongTw> g_tw;       // (device, N, dtype)

int long_twiddles(long long N, int dtype, LongTw* out) {
    int dev = 0;
    DFFT_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mutex);
    auto key = std::make_tuple(dev, N, dtype);
    auto it = g_tw.find(key);
    if (it != g_tw.end()) {
        *out = it->second;
        return DFFT_OK;
    }
    const long long   nhi = (N + kLo - 1) >> kLoBits;
    const long double two_pi = 6.283185307179586476925286766559005768L;
    auto              fill = [&](long long count, long long step, void** dptr) -> int {
        const size_t eb = elem_bytes(dtype);
        std::vector<char> h((size_t)count * eb);
        for (long long t = 0; t < count; ++t) {
            const long double a = two_pi * (long double)((t * step) % N) / (long double)N;
            if (dtype == DFFT_F64) {
                ((double*)h.data())[2 * t] = (double)cosl(a);
                ((double*)h.data())[2 * t + 1] = (double)(-sinl(a));
            } else {
                ((float*)h.data())[2 * t] = (float)cosl(a);
                ((float*)h.data())[2 * t + 1] = (float)(-sinl(a));
            }
        }
        DFFT_HIP_TRY(hipMalloc(dptr, h.size()));
        DFFT_HIP_TRY(hipMemcpy(*dptr, h.data(), h.size(), hipMemcpyHostToDevice));
        return DFFT_OK;
    };
    LongTw t{nullptr, nullptr};
    int    rc = fill(nhi, kLo, &t.hi);
    if (rc == DFFT_OK) rc = fill(kLo, 1, &t.lo);
    if (rc) return rc;
    g_tw[key] = t;
    *out = t;
    return DFFT_OK;
}

int cols_launch(const void* in, void* out, int n, long long width, long long batch, int dtype, int dir, hipStream_t s) {
    const void* tw = nullptr;
    int         rc = get_twiddles(n, dtype, &tw);
    if (rc) return rc;
    FftLaunch L;
    std::memset(&L, 0, sizeof(L));
    L.dtype = dtype;
    L.n = n;
    L.dir = dir;
    L.in = in;
    L.out = out;
    L.tw = tw;
    if (width == 1) {  // contiguous rows
        L.cols = 0;
        L.imap = L.omap = AxisMap{n, 1, 0, 1, 0, 0, 1, 0};
        L.itile = L.otile = TileMap{(long long)n, 0};
        L.ntiles = batch;
        L.tiles_per_a = 1;
        L.ncols = 1;
    } else {
        if (width >= (1ll << 31)) return fail(DFFT_EUNSUPPORTED, "long FFT: more than 2^31 columns");
        L.cols = 1;
        L.imap = L.omap = AxisMap{n, 1, 0, width, 1, 0, 1, 0};
        L.itile = L.otile = TileMap{(long long)n * width, 1};
        L.na = batch;
        L.ncols = (int)width;
    }
    hipError_t e = launch_fft(L, s);
    if (e == hipSuccess) return DFFT_OK;
    return fail(e == hipErrorInvalidValue ? DFFT_EUNSUPPORTED : DFFT_EHIP, std::string("long FFT: ") + hipGetErrorString(e));
}

}  // namespace

bool long_split(long long n, int* n1, int* n2) {
    if (n <= 4096 || n > (1ll << 24)) return false;
    // the most balanced pair of tuned single-pass lengths (the larger factor on the contiguous side)
    int best1 = 0, best2 = 0;
    for (int a = 2; (long long)a * a <= n; ++a) {
        if (n % a) continue;
        const long long b = n / a;
        if (b > 4096 || !fft_length_tuned(a) || !fft_length_tuned((int)b)) continue;
        best1 = a;
        best2 = (int)b;
    }
    if (!best1) return false;
    *n1 = best1;
    *n2 = best2;
    return true;
}

// Scratch for callers without a plan (the 1-D API): one grow-only buffer per (device, stream).  The entry's lock is held from
// long_scratch() until long_scratch_release(), i.e. across the caller's enqueue: a second host thread on the same stream waits
// for the first one's kernels to be queued, and when it has to grow the buffer its hipStreamSynchronize covers them before the
// old buffer is freed.  (hipMallocAsync / hipFreeAsync per call was tried first and gave wrong results on the NULL stream.)
// The lease IS the entry: release needs no look-up (nothing that can fail or find another entry because the calling thread's
// current device changed in between), so a lease that was taken is always returned.
struct ScratchEntry {
    std::mutex m;
    void*      p = nullptr;
    size_t     bytes = 0;
    int        dev = 0;
};
static std::map<std::pair<int, hipStream_t>, std::unique_ptr<ScratchEntry>> g_scratch;

void* long_scratch(size_t bytes, hipStream_t stream, LongScratchLease* lease) {
    if (!lease) return nullptr;
    *lease = nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    ScratchEntry* e = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mutex);
        auto&                       slot = g_scratch[std::make_pair(dev, stream)];
        if (!slot) {
            slot.reset(new ScratchEntry);
            slot->dev = dev;
        }
        e = slot.get();
    }
    e->m.lock();
    if (e->bytes < bytes) {
        if (e->p) {
            (void)hipStreamSynchronize(stream);  // work still using the old buffer
            (void)hipFree(e->p);
        }
        e->p = nullptr;
        e->bytes = 0;
        if (hipMalloc(&e->p, bytes) != hipSuccess) {
            (void)hipGetLastError();
            e->p = nullptr;
            e->m.unlock();
            return nullptr;
        }
        e->bytes = bytes;
    }
    *lease = e;
    return e->p;
}
void long_scratch_release(LongScratchLease lease) {
    if (lease) static_cast<ScratchEntry*>(lease)->m.unlock();
}
// Frees every cached buffer that is not leased.  The cached stream handles belong to callers and may have been destroyed since, so
// they are not touched: the owning device is drained instead, and neither that wait nor the frees run under the registry lock
// (entries are never removed from the registry, so the pointers collected under it stay valid).
void long_scratch_trim() {
    std::vector<ScratchEntry*> entries;
    {
        std::lock_guard<std::mutex> lk(g_mutex);
        for (auto& kv : g_scratch) entries.push_back(kv.second.get());
    }
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (ScratchEntry* e : entries) {
        if (!e->m.try_lock()) continue;  // in use right now
        if (e->p) {
            if (hipSetDevice(e->dev) == hipSuccess) {
                (void)hipDeviceSynchronize();
                (void)hipFree(e->p);
                e->p = nullptr;
                e->bytes = 0;
            } else {
                (void)hipGetLastError();
            }
        }
        e->m.unlock();
    }
    if (have_cur) (void)hipSetDevice(cur);
}

int long_fft(const void* in, void* out, long long n, long long s, long long batch, int dtype, int dir, double scale, void* scratch,
             hipStream_t stream) {
    int N1 = 0, N2 = 0;
    if (!long_split(n, &N1, &N2))
        return fail(DFFT_EUNSUPPORTED, "FFT length " + std::to_string(n) + " is not a product of two tuned single-pass lengths");
    if (batch <= 0 || s <= 0) return DFFT_OK;
    const long long total = batch * n * s;
    if (!scratch) return fail(DFFT_EINVAL, "long FFT: no scratch buffer");
    LongTw tw;
    int    rc = long_twiddles(n, dtype, &tw);
    if (rc) return rc;
    // A: N1-point transforms, stride N2*s, for the N2*s columns of every batch item
    rc = cols_launch(in, scratch, N1, (long long)N2 * s, batch, dtype, dir, stream);
    if (rc) return rc;
    (void)hipGetLastError();
    long long grid = (total + 255) / 256;
    if (grid > 256 * 16) grid = 256 * 16;
    if (dtype == DFFT_F64)
        hipLaunchKernelGGL((long_twiddle_kernel<double2>), dim3((unsigned)grid), dim3(256), 0, stream, (double2*)scratch, total, N1, N2, s,
                           (const double2*)tw.hi, (const double2*)tw.lo, dir);
    else
        hipLaunchKernelGGL((long_twiddle_kernel<float2>), dim3((unsigned)grid), dim3(256), 0, stream, (float2*)scratch, total, N1, N2, s,
                           (const float2*)tw.hi, (const float2*)tw.lo, dir);
    DFFT_HIP_TRY(hipGetLastError());
    // B: N2-point transforms, stride s, for every (batch item, k1)
    rc = cols_launch(scratch, scratch, N2, s, batch * N1, dtype, dir, stream);
    if (rc) return rc;
    // P: natural order (+ scaling)
    const double sc = scale == 0.0 ? 1.0 : scale;
    if (s == 1) {
        const int tiles_r = (N1 + 31) / 32, tiles_c = (N2 + 31) / 32;
        if (batch >= 65536) {  // grid.y limit: go through the runs kernel
            grid = std::min<long long>(256 * 16, (batch * N1 * N2 + 3) / 4);
            if (dtype == DFFT_F64)
                hipLaunchKernelGGL((long_permute_runs_kernel<double2>), dim3((unsigned)grid), dim3(256), 0, stream, (const double2*)scratch,
                                   (double2*)out, batch * N1 * N2, N1, N2, 1ll, sc);
            else
                hipLaunchKernelGGL((long_permute_runs_kernel<float2>), dim3((unsigned)grid), dim3(256), 0, stream, (const float2*)scratch,
                                   (float2*)out, batch * N1 * N2, N1, N2, 1ll, (float)sc);
        } else if (dtype == DFFT_F64) {
            hipLaunchKernelGGL((long_permute_tile_kernel<double2>), dim3((unsigned)(tiles_r * tiles_c), (unsigned)batch), dim3(256), 0, stream,
                               (const double2*)scratch, (double2*)out, N1, N2, tiles_c, sc);
        } else {
            hipLaunchKernelGGL((long_permute_tile_kernel<float2>), dim3((unsigned)(tiles_r * tiles_c), (unsigned)batch), dim3(256), 0, stream,
                               (const float2*)scratch, (float2*)out, N1, N2, tiles_c, (float)sc);
        }
    } else {
        const long long nruns = batch * N1 * N2;
        grid = std::min<long long>(256 * 16, (nruns + 3) / 4);
        if (dtype == DFFT_F64)
            hipLaunchKernelGGL((long_permute_runs_kernel<double2>), dim3((unsigned)grid), dim3(256), 0, stream, (const double2*)scratch,
                               (double2*)out, nruns, N1, N2, s, sc);
        else
            hipLaunchKernelGGL((long_permute_runs_kernel<float2>), dim3((unsigned)grid), dim3(256), 0, stream, (const float2*)scratch,
                               (float2*)out, nruns, N1, N2, s, (float)sc);
    }
    DFFT_HIP_TRY(hipGetLastError());
    return DFFT_OK;
}

}  // namespace dfft
