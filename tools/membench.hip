// tools/membench.hip -- developer measurement: what this GPU gives plain data movement, by working-set size.
//   copy  : out[i] = in[i], 16 B per lane, grid-stride; sizes from 2 GiB (HBM) down to L2-resident, `reps` passes inside one launch
//           for the small sizes (a fixed grid keeps every address on the same XCD from pass to pass)
//   read  : sum of in[], write : out[] = const   (2 GiB and cache-resident)
//   census: XCC_ID of every workgroup of a 256/512/1024-workgroup grid (dispatch placement, for information only)
//   barrier: cost of an XCD-local counter barrier between the workgroups of one XCD (teams by XCC_ID), 1 and 2 per CU
// Reconciles the copy ceiling quoted in MI355X_MICROARCH.md (6.29 TB/s, float4 copy) with what the FFT passes see.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
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

template <int THREADS, int U, bool NTL, bool NTS>
__global__ void __launch_bounds__(THREADS) copy_k(const d2* in, d2* out, size_t n, int reps) {
    const size_t stride = (size_t)gridDim.x * THREADS;
    for (int r = 0; r < reps; ++r) {
        for (size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += stride * U) {
            d2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (i + u * stride < n) v[u] = ld<NTL>(in + i + u * stride);
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (i + u * stride < n) st<NTS>(out + i + u * stride, v[u]);
        }
    }
}
// contiguous 8 KiB per wave per step (what an FFT row kernel does): wave w of the grid moves rows w, w + nwaves, ...
template <int THREADS, bool NTL, bool NTS>
__global__ void __launch_bounds__(THREADS) copy_rows_k(const d2* in, d2* out, size_t nrows, int reps) {
    const int    lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (THREADS / 64) + threadIdx.x / 64, nw = (size_t)gridDim.x * (THREADS / 64);
    for (int r = 0; r < reps; ++r)
        for (size_t row = wave; row < nrows; row += nw) {
            d2 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = ld<NTL>(in + row * 512 + lane + 64 * k);
#pragma unroll
            for (int k = 0; k < 8; ++k) st<NTS>(out + row * 512 + lane + 64 * k, v[k]);
        }
}
template <int THREADS, int U, bool NTL> __global__ void __launch_bounds__(THREADS) read_k(const d2* in, d2* sink, size_t n, int reps) {
    const size_t stride = (size_t)gridDim.x * THREADS;
    d2 acc = {0, 0};
    for (int r = 0; r < reps; ++r)
        for (size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += stride * U) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (i + u * stride < n) acc += ld<NTL>(in + i + u * stride);
        }
    if (acc.x == 123.456) sink[0] = acc;
}
template <int THREADS, bool NTS> __global__ void __launch_bounds__(THREADS) write_k(d2* out, size_t n, int reps) {
    const size_t stride = (size_t)gridDim.x * THREADS;
    for (int r = 0; r < reps; ++r) {
        const d2 v = {(double)r, 1.0};
        for (size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += stride) st<NTS>(out + i, v);
    }
}

__global__ void census_k(unsigned* info) {
    if (threadIdx.x == 0) {
        unsigned x, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        info[blockIdx.x * 2] = x;
        info[blockIdx.x * 2 + 1] = hw;
    }
}

struct alignas(128) Line { unsigned v; unsigned pad[31]; };
struct BarCtl {
    Line registered, error;
    Line xcc_count[16];
    Line cnt[16], gen[16];
};
__device__ __forceinline__ unsigned ldu(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ bool bwait(const unsigned* w, unsigned target, BarCtl* c) {
    const unsigned long long t0 = wall_clock64();
    while (ldu(w) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (ldu(&c->error.v)) return false;
        if (wall_clock64() - t0 > 2000000ull) {  // 20 ms
            __hip_atomic_store(&c->error.v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}
// `iters` XCD-local barriers between the workgroups sharing an XCC_ID (global == 1: one barrier over the whole grid instead)
__global__ void __launch_bounds__(512) barrier_k(BarCtl* c, int iters, int global) {
    __shared__ unsigned sh[4];
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        x &= 15u;
        if (global) x = 0;
        __hip_atomic_fetch_add(&c->xcc_count[x].v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&c->registered.v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = bwait(&c->registered.v, gridDim.x, c);
        sh[0] = x;
        sh[1] = ldu(&c->xcc_count[x].v);
        sh[2] = ok;
    }
    __syncthreads();
    if (!sh[2]) return;
    const unsigned x = sh[0], T = sh[1];
    for (int it = 1; it <= iters; ++it) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned old = __hip_atomic_fetch_add(&c->cnt[x].v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1u == (unsigned)it * T) __hip_atomic_store(&c->gen[x].v, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh[2] = bwait(&c->gen[x].v, (unsigned)it, c);
        }
        __syncthreads();
        if (!sh[2]) return;
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
    const size_t nmax = 512ull * 512 * 512;  // 2 GiB of 16-byte elements
    d2 *a, *b;
    CK(hipMalloc(&a, nmax * 16));
    CK(hipMalloc(&b, nmax * 16));
    CK(hipMemset(a, 1, nmax * 16));
    CK(hipMemset(b, 0, nmax * 16));
    CK(hipStreamCreateWithFlags(&g_s, hipStreamNonBlocking));

    struct Sz { const char* name; size_t n; int reps; };
    const Sz sizes[] = {{"2 GiB", nmax, 1},          {"512 MiB", nmax / 4, 4},    {"128 MiB", nmax / 16, 16},
                        {"64 MiB", nmax / 32, 32},   {"32 MiB", nmax / 64, 64},   {"8 MiB", nmax / 256, 256},
                        {"2 MiB", nmax / 1024, 1024}};
    printf("== copy (read n + write n bytes; rate counts both), 16 B per lane ==\n");
    for (const Sz& z : sizes) {
        auto rep = [&](const char* what, int grid, double ms) {
            printf("copy %-8s %-26s grid %5d  %8.3f ms  %7.0f GB/s\n", z.name, what, grid, ms, 2.0 * z.n * 16 * z.reps / ms * 1e-6);
            fflush(stdout);
        };
        for (int grid : {512, 768, 1024, 2048, 4096}) {
            rep("256thr U1 plain", grid, time_ms([&] { hipLaunchKernelGGL((copy_k<256, 1, false, false>), dim3(grid), dim3(256), 0, g_s, a, b, z.n, z.reps); }));
            rep("256thr U1 nt", grid, time_ms([&] { hipLaunchKernelGGL((copy_k<256, 1, true, true>), dim3(grid), dim3(256), 0, g_s, a, b, z.n, z.reps); }));
            rep("256thr U4 nt", grid, time_ms([&] { hipLaunchKernelGGL((copy_k<256, 4, true, true>), dim3(grid), dim3(256), 0, g_s, a, b, z.n, z.reps); }));
        }
        for (int grid : {256, 512, 768}) {
            rep("512thr U1 nt", grid, time_ms([&] { hipLaunchKernelGGL((copy_k<512, 1, true, true>), dim3(grid), dim3(512), 0, g_s, a, b, z.n, z.reps); }));
            rep("512thr U8 nt", grid, time_ms([&] { hipLaunchKernelGGL((copy_k<512, 8, true, true>), dim3(grid), dim3(512), 0, g_s, a, b, z.n, z.reps); }));
            rep("512thr U8 plain", grid, time_ms([&] { hipLaunchKernelGGL((copy_k<512, 8, false, false>), dim3(grid), dim3(512), 0, g_s, a, b, z.n, z.reps); }));
        }
        for (int grid : {512, 1024}) {
            rep("rows 8KiB/wave 256thr nt", grid, time_ms([&] { hipLaunchKernelGGL((copy_rows_k<256, true, true>), dim3(grid), dim3(256), 0, g_s, a, b, z.n / 512, z.reps); }));
            rep("rows 8KiB/wave 256thr plain", grid, time_ms([&] { hipLaunchKernelGGL((copy_rows_k<256, false, false>), dim3(grid), dim3(256), 0, g_s, a, b, z.n / 512, z.reps); }));
        }
    }
    printf("== hipMemcpyAsync DtoD 2 GiB ==\n");
    {
        double ms = time_ms([&] { CK(hipMemcpyAsync(b, a, nmax * 16, hipMemcpyDeviceToDevice, g_s)); });
        printf("memcpy 2 GiB  %8.3f ms  %7.0f GB/s\n", ms, 2.0 * nmax * 16 / ms * 1e-6);
    }
    printf("== read only / write only ==\n");
    for (const Sz& z : sizes) {
        for (int grid : {1024, 2048}) {
            double ms = time_ms([&] { hipLaunchKernelGGL((read_k<256, 4, false>), dim3(grid), dim3(256), 0, g_s, a, b, z.n, z.reps); });
            printf("read  %-8s 256thr U4 plain grid %5d  %8.3f ms  %7.0f GB/s\n", z.name, grid, ms, 1.0 * z.n * 16 * z.reps / ms * 1e-6);
            ms = time_ms([&] { hipLaunchKernelGGL((read_k<256, 4, true>), dim3(grid), dim3(256), 0, g_s, a, b, z.n, z.reps); });
            printf("read  %-8s 256thr U4 nt    grid %5d  %8.3f ms  %7.0f GB/s\n", z.name, grid, ms, 1.0 * z.n * 16 * z.reps / ms * 1e-6);
            ms = time_ms([&] { hipLaunchKernelGGL((write_k<256, false>), dim3(grid), dim3(256), 0, g_s, b, z.n, z.reps); });
            printf("write %-8s 256thr plain    grid %5d  %8.3f ms  %7.0f GB/s\n", z.name, grid, ms, 1.0 * z.n * 16 * z.reps / ms * 1e-6);
            ms = time_ms([&] { hipLaunchKernelGGL((write_k<256, true>), dim3(grid), dim3(256), 0, g_s, b, z.n, z.reps); });
            printf("write %-8s 256thr nt       grid %5d  %8.3f ms  %7.0f GB/s\n", z.name, grid, ms, 1.0 * z.n * 16 * z.reps / ms * 1e-6);
            fflush(stdout);
        }
    }
    printf("== workgroup -> XCC census ==\n");
    {
        unsigned* info;
        CK(hipMalloc(&info, 4096 * 8));
        for (int grid : {256, 512, 1024}) {
            hipLaunchKernelGGL(census_k, dim3(grid), dim3(512), 0, g_s, info);
            CK(hipStreamSynchronize(g_s));
            std::vector<unsigned> h(grid * 2);
            CK(hipMemcpy(h.data(), info, grid * 8, hipMemcpyDeviceToHost));
            int per[16] = {0}, mism = 0;
            for (int i = 0; i < grid; ++i) {
                per[h[2 * i] & 15]++;
                mism += (int)(h[2 * i] & 15) != i % 8;
            }
            printf("grid %4d: blocks per XCC:", grid);
            for (int x = 0; x < 16; ++x)
                if (per[x]) printf(" %d", per[x]);
            printf("   blocks with XCC_ID != blockIdx %% 8: %d\n", mism);
        }
    }
    printf("== XCD-local barrier (teams by XCC_ID) ==\n");
    {
        BarCtl* c;
        CK(hipMalloc(&c, sizeof(BarCtl)));
        for (int global : {0, 1})
            for (int grid : {256, 512}) {
                const int iters = 2000;
                double ms = time_ms([&] {
                    CK(hipMemsetAsync(c, 0, sizeof(BarCtl), g_s));
                    hipLaunchKernelGGL(barrier_k, dim3(grid), dim3(512), 0, g_s, c, iters, global);
                });
                double ms0 = time_ms([&] {
                    CK(hipMemsetAsync(c, 0, sizeof(BarCtl), g_s));
                    hipLaunchKernelGGL(barrier_k, dim3(grid), dim3(512), 0, g_s, c, 0, global);
                });
                BarCtl h;
                CK(hipMemcpy(&h, c, sizeof(h), hipMemcpyDeviceToHost));
                printf("%s barrier, grid %d: %.3f us per barrier (launch alone %.1f us, error flag %u)\n", global ? "grid-wide" : "XCD-local", grid,
                       (ms - ms0) * 1e3 / iters, ms0 * 1e3, h.error.v);
            }
    }
    return 0;
}
