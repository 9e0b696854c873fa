// tools/kbench9.hip -- developer experiment: what does this GPU give a kernel that only MOVES 2 GiB -> 2 GiB with the
// access patterns of the three FFT passes (no LDS, no math)?  Separates "pattern-bound" from "kernel-bound".
//   linear      : thread i copies element i (16 B), grid-stride                      (Z pass pattern)
//   ytile       : tile = 512 rows x 128 B at stride 8 KiB, written in place pattern  (Y pass pattern)
//   xtile       : tile = 512 rows x 128 B at stride 4 MiB, written as one 64 KiB run (X pass pattern, staged store)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
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

// U elements per thread per iteration, all loads issued before the stores
template <int U, bool NTL, bool NTS> __global__ void __launch_bounds__(512) linear_copy(const d2* in, d2* out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride * U) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * stride < n) v[u] = ld<NTL>(in + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * stride < n) st<NTS>(out + i + u * stride, v[u]);
    }
}

// tile (a, b): 512 rows (idx) x 8 columns; thread (j = tid / 8, c = tid % 8) moves rows j + 64 k, k < 8.
// in : base a*ia + b*8 + idx*istride + c       out (same == true): same pattern with oa / ostride
// out (transposed run): base a*oa + b*8*512 + lin,  lin = tid + 512 k   (what the staged store writes)
template <bool RUN, bool NTL, bool NTS, bool PF>
__global__ void __launch_bounds__(512) tile_copy(const d2* in, d2* out, long long ia, long long istride, long long oa,
                                                 long long ostride, unsigned ntiles, unsigned tiles_per_a) {
    const int tid = threadIdx.x, c = tid & 7, j = tid >> 3;
    d2 v[8], w[8];
    auto load = [&](unsigned t, d2* dst) {
        const unsigned a = t / tiles_per_a, b = t - a * tiles_per_a;
        const d2* ip = in + (long long)a * ia + (long long)b * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[k] = ld<NTL>(ip + (long long)(j + 64 * k) * istride + c);
    };
    if (PF && blockIdx.x < ntiles) load(blockIdx.x, v);
    for (unsigned t = blockIdx.x; t < ntiles; t += gridDim.x) {
        if (PF) {
            if (t + gridDim.x < ntiles) load(t + gridDim.x, w);
        } else {
            load(t, v);
        }
        const unsigned a = t / tiles_per_a, b = t - a * tiles_per_a;
        if (RUN) {
            d2* op = out + (long long)a * oa + (long long)b * 8 * 512;
#pragma unroll
            for (int k = 0; k < 8; ++k) st<NTS>(op + tid + 512 * k, v[k]);
        } else {
            d2* op = out + (long long)a * oa + (long long)b * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) st<NTS>(op + (long long)(j + 64 * k) * ostride + c, v[k]);
        }
        if (PF) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = w[k];
        }
    }
}

// same as tile_copy<RUN=true, nt, prefetch> but every XCD (blockIdx % 8) walks its own contiguous eighth of the tiles
template <bool NTL, bool NTS>
__global__ void __launch_bounds__(512) xtile_xcd(const d2* in, d2* out, long long ia, long long istride, long long oa,
                                                 unsigned ntiles, unsigned tiles_per_a) {
    const int tid = threadIdx.x, c = tid & 7, j = tid >> 3;
    const unsigned xcd = blockIdx.x & 7, local = blockIdx.x >> 3, per = ntiles / 8, step = gridDim.x >> 3;
    for (unsigned tl = local; tl < per; tl += step) {
        const unsigned t = xcd * per + tl;
        const unsigned a = t / tiles_per_a, b = t - a * tiles_per_a;
        const d2* ip = in + (long long)a * ia + (long long)b * 8;
        d2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = ld<NTL>(ip + (long long)(j + 64 * k) * istride + c);
        d2* op = out + (long long)a * oa + (long long)b * 8 * 512;
#pragma unroll
        for (int k = 0; k < 8; ++k) st<NTS>(op + tid + 512 * k, v[k]);
    }
}

template <int THREADS, bool NTL, bool NTS> __global__ void __launch_bounds__(THREADS) linear_copy_t(const d2* in, d2* out, size_t n) {
    const size_t stride = (size_t)gridDim.x * THREADS;
    for (size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += stride) st<NTS>(out + i, ld<NTL>(in + i));
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 7;
    const size_t n = 512ull * 512 * 512;
    d2 *a, *b;
    CK(hipMalloc(&a, n * 16));
    CK(hipMalloc(&b, n * 16));
    CK(hipMemset(a, 1, n * 16));
    CK(hipMemset(b, 0, n * 16));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct Case { std::string name; std::function<void()> run; };
    std::vector<Case> cases;
    for (int bpc : {2, 4, 8}) {
        const int grid = 256 * bpc;
        cases.push_back({"linear U1 plain      grid " + std::to_string(grid), [=] { linear_copy<1, false, false><<<grid, 512, 0, s>>>(a, b, n); }});
        cases.push_back({"linear U4 plain      grid " + std::to_string(grid), [=] { linear_copy<4, false, false><<<grid, 512, 0, s>>>(a, b, n); }});
        cases.push_back({"linear U4 nt         grid " + std::to_string(grid), [=] { linear_copy<4, true, true><<<grid, 512, 0, s>>>(a, b, n); }});
        cases.push_back({"linear U8 nt         grid " + std::to_string(grid), [=] { linear_copy<8, true, true><<<grid, 512, 0, s>>>(a, b, n); }});
    }
    for (int grid : {128, 256, 384, 512, 768}) {
        cases.push_back({"linear U1 plain 512thr grid " + std::to_string(grid), [=] { linear_copy_t<512, false, false><<<grid, 512, 0, s>>>(a, b, n); }});
        cases.push_back({"linear U1 nt    512thr grid " + std::to_string(grid), [=] { linear_copy_t<512, true, true><<<grid, 512, 0, s>>>(a, b, n); }});
        cases.push_back({"linear U1 plain 256thr grid " + std::to_string(grid * 2), [=] { linear_copy_t<256, false, false><<<grid * 2, 256, 0, s>>>(a, b, n); }});
        cases.push_back({"linear U1 plain 1024thr grid " + std::to_string(grid / 2), [=] { linear_copy_t<1024, false, false><<<grid / 2, 1024, 0, s>>>(a, b, n); }});
    }
    for (int grid : {256, 512, 1024}) {
        cases.push_back({"xtile xcd-contiguous nt grid " + std::to_string(grid), [=] { xtile_xcd<true, true><<<grid, 512, 0, s>>>(a, b, 512, 512 * 512, 512 * 512, 512 * 64, 64); }});
        cases.push_back({"xtile xcd-contiguous plain grid " + std::to_string(grid), [=] { xtile_xcd<false, false><<<grid, 512, 0, s>>>(a, b, 512, 512 * 512, 512 * 512, 512 * 64, 64); }});
    }
    cases.push_back({"hipMemcpyAsync DtoD", [=] { CK(hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, s)); }});
    const unsigned nt = 512 * 64;  // 512 `a` slices x 64 column tiles
    for (int bpc : {1, 2, 4}) {
        const int grid = 256 * bpc;
        const std::string g = "  grid " + std::to_string(grid);
        // Y pattern: a = x plane (stride 512*512), rows at stride 512
        cases.push_back({"ytile plain" + g, [=] { tile_copy<false, false, false, false><<<grid, 512, 0, s>>>(a, b, 512 * 512, 512, 512 * 512, 512, nt, 64); }});
        cases.push_back({"ytile prefetch" + g, [=] { tile_copy<false, false, false, true><<<grid, 512, 0, s>>>(a, b, 512 * 512, 512, 512 * 512, 512, nt, 64); }});
        cases.push_back({"ytile prefetch nt" + g, [=] { tile_copy<false, true, true, true><<<grid, 512, 0, s>>>(a, b, 512 * 512, 512, 512 * 512, 512, nt, 64); }});
        // X pattern: a = y (stride 512), rows (x) at stride 512*512; out [y][z][kx]: a stride 512*512, tile run 8*512
        cases.push_back({"xtile plain" + g, [=] { tile_copy<true, false, false, false><<<grid, 512, 0, s>>>(a, b, 512, 512 * 512, 512 * 512, 0, nt, 64); }});
        cases.push_back({"xtile prefetch" + g, [=] { tile_copy<true, false, false, true><<<grid, 512, 0, s>>>(a, b, 512, 512 * 512, 512 * 512, 0, nt, 64); }});
        cases.push_back({"xtile prefetch nt" + g, [=] { tile_copy<true, true, true, true><<<grid, 512, 0, s>>>(a, b, 512, 512 * 512, 512 * 512, 0, nt, 64); }});
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<std::vector<float>> ms(cases.size());
    for (int r = 0; r < rounds + 1; ++r)
        for (size_t i = 0; i < cases.size(); ++i) {
            CK(hipEventRecord(e0, s));
            cases[i].run();
            CK(hipGetLastError());
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float t;
            CK(hipEventElapsedTime(&t, e0, e1));
            if (r > 0) ms[i].push_back(t);
        }
    for (size_t i = 0; i < cases.size(); ++i) {
        auto v = ms[i];
        std::sort(v.begin(), v.end());
        printf("%-36s median %.3f ms  %.0f GB/s\n", cases[i].name.c_str(), v[v.size() / 2], 2.0 * n * 16 / v[v.size() / 2] / 1e6);
    }
    return 0;
}
