// dfft_kernels.hip -- length dispatcher + the un-fused reorder kernels (pack / tile transpose / scale).
//
// The fused pipeline never launches pack/transpose (the FFT kernels' address maps do that work); they exist so the
// reference's stage structure (t1 = pack, t3 = transpose + FFT) can be reproduced and timed stage by stage
// (DFFT_FUSE=0), and as independent checks of the fused address maps.
#include <mutex>

#include "dfft_kernels.h"
#include "dfft_plans.h"

namespace dfft {

template <int N> hipError_t launch_n(const FftLaunch& L, hipStream_t stream);

#define DFFT_EXTERN_PLAN(N, GRP, E, ...) extern template hipError_t launch_n<N>(const FftLaunch&, hipStream_t);
DFFT_PLAN_TABLE(DFFT_EXTERN_PLAN)
#undef DFFT_EXTERN_PLAN

bool fft_length_supported(int n) {
    switch (n) {
#define DFFT_CASE(N, GRP, E, ...) case N:
        DFFT_PLAN_TABLE(DFFT_CASE)
#undef DFFT_CASE
        return true;
        default: return generic_length_supported(n);
    }
}

bool fft_length_tuned(int n) {
    switch (n) {
#define DFFT_CASE(N, GRP, E, ...) case N:
        DFFT_PLAN_TABLE(DFFT_CASE)
#undef DFFT_CASE
        return true;
        default: return false;
    }
}

hipError_t launch_fft(const FftLaunch& L, hipStream_t stream) {
    if ((L.cols ? L.na : L.ntiles) <= 0) return hipSuccess;
    switch (L.n) {
#define DFFT_CASE(N, GRP, E, ...) \
    case N: return launch_n<N>(L, stream);
        DFFT_PLAN_TABLE(DFFT_CASE)
#undef DFFT_CASE
        default: return launch_generic(L, stream);  // 7-smooth lengths without a tuned plan (dfft_generic.hip)
    }
}

// ---------------------------------------------------------------------------------------------------------------
// pack / unpack: [xl][N1][N2] <-> [d][xl][yl_d][N2]  (index map: SURVEY Appendix B, kernel_func.cpp:73-100).
// Pure row copy: one wave moves one z-row with 16-byte accesses; rows are grid-strided.
template <class V, int DIR>
__global__ void __launch_bounds__(256) pack_rows_kernel(const V* in, V* out, int xl, int n1, int n2, int yl, int ylast,
                                                        int P) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long nrows = (long long)xl * n1;
    for (long long row = (long long)blockIdx.x * 4 + wave; row < nrows; row += (long long)gridDim.x * 4) {
        const int xi = (int)(row / n1), y = (int)(row - (long long)xi * n1);
        int d = y / yl;
        if (d > P - 1) d = P - 1;
        const int yy = y - d * yl;
        const int yw = (d == P - 1) ? ylast : yl;
        const long long nat = row * n2;
        const long long pk = ((long long)d * xl * yl + (long long)xi * yw + yy) * n2;
        const V* src = DIR > 0 ? in + nat : in + pk;
        V* dst = DIR > 0 ? out + pk : out + nat;
        for (int z = lane; z < n2; z += 64) dst[z] = src[z];
    }
}

hipError_t launch_pack(int dtype, int dir, const void* in, void* out, int xl, int n1, int n2, int yl, int ylast, int P,
                       hipStream_t stream) {
    const long long nrows = (long long)xl * n1;
    if (nrows <= 0) return hipSuccess;
    (void)hipGetLastError();
    long long grid = (nrows + 3) / 4;
    if (grid > 256 * 8) grid = 256 * 8;
    if (dtype == F64) {
        if (dir > 0) hipLaunchKernelGGL((pack_rows_kernel<double2, +1>), dim3((unsigned)grid), dim3(256), 0, stream, (const double2*)in, (double2*)out, xl, n1, n2, yl, ylast, P);
        else hipLaunchKernelGGL((pack_rows_kernel<double2, -1>), dim3((unsigned)grid), dim3(256), 0, stream, (const double2*)in, (double2*)out, xl, n1, n2, yl, ylast, P);
    } else if (dtype == F32) {
        if (dir > 0) hipLaunchKernelGGL((pack_rows_kernel<float2, +1>), dim3((unsigned)grid), dim3(256), 0, stream, (const float2*)in, (float2*)out, xl, n1, n2, yl, ylast, P);
        else hipLaunchKernelGGL((pack_rows_kernel<float2, -1>), dim3((unsigned)grid), dim3(256), 0, stream, (const float2*)in, (float2*)out, xl, n1, n2, yl, ylast, P);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// 2D transpose out[c][r] = in[r][c] of a rows x cols matrix through a padded LDS tile.
// Tile = 64 x 64 elements (wave64-wide: every global access of a wave is 64 consecutive elements = 1 KiB fp64),
// LDS row pitch 65 elements so the transposed (column) reads hit distinct banks.  Replaces the reference's 16 x 16
// (+1) tiles (fast_transpose.h:11, kernels_201.cpp:28-70), whose 256-byte rows under-fill a wave64 access.
template <class V> __global__ void __launch_bounds__(256) transpose_tile_kernel(const V* in, V* out, long long rows, long long cols, long long tiles_c) {
    extern __shared__ __attribute__((aligned(16))) char dfft_tr_smem[];
    V (*tile)[65] = reinterpret_cast<V (*)[65]>(dfft_tr_smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tr = blockIdx.x / tiles_c, tc = blockIdx.x - tr * tiles_c;
    const long long r0 = tr * 64, c0 = tc * 64;
#pragma unroll 4
    for (int i = wave; i < 64; i += 4) {
        const long long r = r0 + i, c = c0 + lane;
        if (r < rows && c < cols) tile[i][lane] = in[r * cols + c];
    }
    __syncthreads();
#pragma unroll 4
    for (int i = wave; i < 64; i += 4) {
        const long long c = c0 + i, r = r0 + lane;
        if (r < rows && c < cols) out[c * rows + r] = tile[lane][i];
    }
}

hipError_t launch_transpose(int dtype, const void* in, void* out, long long rows, long long cols, hipStream_t stream) {
    if (rows <= 0 || cols <= 0) return hipSuccess;
    const long long tiles_r = (rows + 63) / 64, tiles_c = (cols + 63) / 64;
    const long long grid = tiles_r * tiles_c;
    if (grid >= (1ll << 31)) return hipErrorInvalidValue;
    (void)hipGetLastError();
    if (dtype == F64) {
        constexpr int lds = 64 * 65 * (int)sizeof(double2);  // 66,560 B > the 64 KiB default cap
        // function attributes are per device: with the thread-per-GPU communicator every device needs its own opt-in to
        // more than 64 KiB of dynamic LDS (one flag per device, set under a lock; the call itself is cheap and idempotent)
        static std::mutex    attr_mutex;
        static bool          attr_set[64] = {false};
        int                  dev = 0;
        hipError_t           e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        {
            std::lock_guard<std::mutex> lk(attr_mutex);
            if (!attr_set[dev]) {
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(transpose_tile_kernel<double2>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                if (e != hipSuccess) return e;
                attr_set[dev] = true;
            }
        }
        hipLaunchKernelGGL((transpose_tile_kernel<double2>), dim3((unsigned)grid), dim3(256), lds, stream, (const double2*)in, (double2*)out, rows, cols, tiles_c);
    } else if (dtype == F32) {
        constexpr int lds = 64 * 65 * (int)sizeof(float2);
        hipLaunchKernelGGL((transpose_tile_kernel<float2>), dim3((unsigned)grid), dim3(256), lds, stream, (const float2*)in, (float2*)out, rows, cols, tiles_c);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <class V, class R> __global__ void __launch_bounds__(256) scale_kernel(V* data, long long count, R s) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
        V v = data[i];
        v.x *= s;
        v.y *= s;
        data[i] = v;
    }
}

hipError_t launch_scale(int dtype, void* data, long long count, double s, hipStream_t stream) {
    if (count <= 0) return hipSuccess;
    (void)hipGetLastError();
    long long grid = (count + 255) / 256;
    if (grid > 256 * 8) grid = 256 * 8;
    if (dtype == F64) hipLaunchKernelGGL((scale_kernel<double2, double>), dim3((unsigned)grid), dim3(256), 0, stream, (double2*)data, count, s);
    else if (dtype == F32) hipLaunchKernelGGL((scale_kernel<float2, float>), dim3((unsigned)grid), dim3(256), 0, stream, (float2*)data, count, (float)s);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace dfft
