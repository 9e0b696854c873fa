// tools/membench2.hip -- developer measurement: is the 4.5 TB/s HBM write ceiling (tools/membench.hip) a property of the
// access pattern?  Write-only and copy kernels over 2 GiB with different shapes of the write stream, plus the runtime's own
// fill, and a read/write mix (some workgroups only read, the others only write) to see how far the two directions overlap.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(...)                                                                           \
    do {                                                                                  \
        hipError_t e_ = (__VA_ARGS__);                                                    \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));
template <bool NT> __device__ __forceinline__ d2 ld(const d2* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(d2* p, d2 v) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// grid-stride, one 16-byte store per lane per step (the membench pattern)
template <int THREADS, bool NT> __global__ void __launch_bounds__(THREADS) w_stride(d2* out, size_t n) {
    const size_t stride = (size_t)gridDim.x * THREADS;
    const d2 v = {1.0, 2.0};
    for (size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += stride) st<NT>(out + i, v);
}
// every workgroup owns one contiguous region and walks it front to back
template <int THREADS, bool NT> __global__ void __launch_bounds__(THREADS) w_blockchunk(d2* out, size_t n) {
    const size_t per = n / gridDim.x, base = per * blockIdx.x;
    const d2 v = {1.0, 2.0};
    for (size_t i = threadIdx.x; i < per; i += THREADS) st<NT>(out + base + i, v);
}
// every wave writes RUN consecutive KiB per step (RUN stores of 1 KiB back to back), waves grid-strided
template <int THREADS, int RUN, bool NT> __global__ void __launch_bounds__(THREADS) w_waverun(d2* out, size_t n) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (THREADS / 64) + threadIdx.x / 64, nw = (size_t)gridDim.x * (THREADS / 64);
    const d2 v = {1.0, 2.0};
    for (size_t c = wave; c * (64 * RUN) < n; c += nw) {
#pragma unroll
        for (int r = 0; r < RUN; ++r) st<NT>(out + c * (64 * RUN) + r * 64 + lane, v);
    }
}
// 64 bytes per lane (4 consecutive 16-byte stores per lane)
template <int THREADS, bool NT> __global__ void __launch_bounds__(THREADS) w_lane64(d2* out, size_t n) {
    const size_t stride = (size_t)gridDim.x * THREADS * 4;
    const d2 v = {1.0, 2.0};
    for (size_t i = ((size_t)blockIdx.x * THREADS + threadIdx.x) * 4; i < n; i += stride) {
#pragma unroll
        for (int r = 0; r < 4; ++r) st<NT>(out + i + r, v);
    }
}
// copy, out displaced by `shift` elements (bank / channel aliasing between the read and the write stream)
template <int THREADS, bool NT> __global__ void __launch_bounds__(THREADS) copy_k(const d2* in, d2* out, size_t n) {
    const size_t stride = (size_t)gridDim.x * THREADS;
    for (size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += stride) st<NT>(out + i, ld<NT>(in + i));
}
// read/write mix: workgroups with (blockIdx % den) < num only read, the others only write; both sweep their buffer completely
template <int THREADS> __global__ void __launch_bounds__(THREADS) mix_k(const d2* in, d2* out, size_t n, int num, int den, d2* sink) {
    const int  role = (int)(blockIdx.x % den) < num;  // 1 = reader
    const int  per = role ? num : den - num;
    const int  idx = role ? (int)(blockIdx.x % den) : (int)(blockIdx.x % den) - num;
    const size_t nb = (size_t)(gridDim.x / den) * per, b = (size_t)(blockIdx.x / den) * per + idx;
    const size_t stride = nb * THREADS;
    if (role) {
        d2 acc = {0, 0};
        for (size_t i = b * THREADS + threadIdx.x; i < n; i += stride) acc += ld<true>(in + i);
        if (acc.x == 123.456) sink[0] = acc;
    } else {
        const d2 v = {1.0, 2.0};
        for (size_t i = b * THREADS + threadIdx.x; i < n; i += stride) st<true>(out + i, v);
    }
}

static hipStream_t g_s;
template <class F> static double time_ms(F&& f, int rounds = 7) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> ms;
    f();
    for (int r = 0; r < rounds; ++r) {
        CK(hipEventRecord(e0, g_s));
        f();
        CK(hipEventRecord(e1, g_s));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t);
    }
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main() {
    const size_t n = 512ull * 512 * 512;
    d2 *a, *b;
    CK(hipMalloc(&a, n * 16));
    CK(hipMalloc(&b, n * 16 + (64 << 20)));
    CK(hipMemset(a, 1, n * 16));
    CK(hipMemset(b, 0, n * 16));
    CK(hipStreamCreateWithFlags(&g_s, hipStreamNonBlocking));
    const double GB = n * 16 * 1e-9;
    auto rep = [&](const char* what, int grid, double ms, double gb) {
        printf("%-44s grid %5d  %8.3f ms  %7.0f GB/s\n", what, grid, ms, gb / ms * 1e3);
        fflush(stdout);
    };
    printf("== write 2 GiB ==\n");
    rep("hipMemsetAsync (D8)", 0, time_ms([&] { CK(hipMemsetAsync(b, 3, n * 16, g_s)); }), GB);
    rep("hipMemsetD32Async", 0, time_ms([&] { CK(hipMemsetD32Async((hipDeviceptr_t)b, 7, n * 4, g_s)); }), GB);
    for (int grid : {128, 256, 512, 1024, 2048, 4096, 16384}) {
        rep("stride 256thr plain", grid, time_ms([&] { hipLaunchKernelGGL((w_stride<256, false>), dim3(grid), dim3(256), 0, g_s, b, n); }), GB);
        rep("stride 256thr nt", grid, time_ms([&] { hipLaunchKernelGGL((w_stride<256, true>), dim3(grid), dim3(256), 0, g_s, b, n); }), GB);
        rep("stride 64thr plain", grid * 4, time_ms([&] { hipLaunchKernelGGL((w_stride<64, false>), dim3(grid * 4), dim3(64), 0, g_s, b, n); }), GB);
        rep("block-contiguous 256thr plain", grid, time_ms([&] { hipLaunchKernelGGL((w_blockchunk<256, false>), dim3(grid), dim3(256), 0, g_s, b, n); }), GB);
        rep("block-contiguous 256thr nt", grid, time_ms([&] { hipLaunchKernelGGL((w_blockchunk<256, true>), dim3(grid), dim3(256), 0, g_s, b, n); }), GB);
        rep("wave runs of 4 KiB plain", grid, time_ms([&] { hipLaunchKernelGGL((w_waverun<256, 4, false>), dim3(grid), dim3(256), 0, g_s, b, n); }), GB);
        rep("wave runs of 8 KiB nt", grid, time_ms([&] { hipLaunchKernelGGL((w_waverun<256, 8, true>), dim3(grid), dim3(256), 0, g_s, b, n); }), GB);
        rep("wave runs of 16 KiB plain", grid, time_ms([&] { hipLaunchKernelGGL((w_waverun<256, 16, false>), dim3(grid), dim3(256), 0, g_s, b, n); }), GB);
        rep("64 B per lane plain", grid, time_ms([&] { hipLaunchKernelGGL((w_lane64<256, false>), dim3(grid), dim3(256), 0, g_s, b, n); }), GB);
    }
    printf("== copy 2 GiB -> 2 GiB, output displaced ==\n");
    for (size_t shift : {(size_t)0, (size_t)256, (size_t)(4096 / 16), (size_t)(65536 / 16), (size_t)((1 << 20) / 16), (size_t)((3 << 20) / 16 + 8),
                         (size_t)((32 << 20) / 16)}) {
        char nm[64];
        snprintf(nm, sizeof nm, "copy nt, out + %zu B", shift * 16);
        rep(nm, 1024, time_ms([&] { hipLaunchKernelGGL((copy_k<256, true>), dim3(1024), dim3(256), 0, g_s, a, b + shift, n); }), 2 * GB);
    }
    printf("== read 2 GiB and write 2 GiB by different workgroups of one launch (rate counts both) ==\n");
    for (int grid : {1024, 2048, 4096})
        for (auto nd : {std::pair<int, int>{1, 2}, {1, 3}, {2, 5}, {1, 4}}) {
            char nm[64];
            snprintf(nm, sizeof nm, "mix: %d of %d workgroups read", nd.first, nd.second);
            const int g = grid / nd.second * nd.second;
            rep(nm, g, time_ms([&] { hipLaunchKernelGGL((mix_k<256>), dim3(g), dim3(256), 0, g_s, a, b, n, nd.first, nd.second, b); }), 2 * GB);
        }
    return 0;
}
