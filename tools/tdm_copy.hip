// tools/tdm_copy.hip -- developer probe (not part of the library): does the memory system move a 2 GiB -> 2 GiB copy faster when
// the whole chip alternates between READ windows and WRITE windows (time-division by the constant-rate wall clock, no
// communication between workgroups) than when every workgroup issues its loads and stores whenever it is ready?
// Round 2 measured read-only 7.0 TB/s, write-only 6.5 TB/s, copy 5.95 TB/s at best: if the mixing of directions at the DRAM is
// what costs the difference, windows of pure traffic should recover part of it (the X pass of the 3D FFT is such a copy).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tdm_copy.hip -o tools/bin/tdm_copy
//   tdm_copy [GiB, default 2]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

typedef double d2v __attribute__((ext_vector_type(2)));

// One persistent workgroup per CU; a tile = THREADS x E x 16 bytes, contiguous.  Tiles are dealt round-robin.
// mode 0: free running, next tile prefetched before the current one is stored (the FFT passes' structure)
// mode 1: loads are issued only inside read windows, stores only inside write windows of a common clock
//         (period ticks of the 100 MHz wall clock, the first rfrac/256 of a period is the read window)
// mode 2: only the stores are gated (write window), loads free
// DEPTH tiles are loaded per window (register budget: DEPTH x E x 4 VGPRs)
template <int THREADS, int E, int DEPTH>
__global__ void __launch_bounds__(THREADS) tdm_kernel(const d2v* __restrict__ a, d2v* __restrict__ b, long long ntiles, int mode, unsigned period,
                                                       unsigned rfrac, unsigned long long* spins) {
    const int       tid = threadIdx.x;
    constexpr int   TILE = THREADS * E;
    const unsigned  rwin = (unsigned)(((unsigned long long)period * rfrac) >> 8);
    unsigned long long waited = 0;
    auto in_read = [&]() { return (unsigned)(wall_clock64() % period) < rwin; };
    auto wait_read = [&]() {
        if (mode != 1) return;
        while (!in_read()) {
            __builtin_amdgcn_s_sleep(1);
            ++waited;
        }
    };
    auto wait_write = [&]() {
        if (mode == 0) return;
        while (in_read()) {
            __builtin_amdgcn_s_sleep(1);
            ++waited;
        }
    };
    d2v v[DEPTH][E];
    for (long long t0 = (long long)blockIdx.x * DEPTH; t0 < ntiles; t0 += (long long)gridDim.x * DEPTH) {
        wait_read();
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const long long t = t0 + d;
            if (t < ntiles) {
#pragma unroll
                for (int k = 0; k < E; ++k) v[d][k] = __builtin_nontemporal_load(a + t * TILE + k * THREADS + tid);
            }
        }
        // the loads must have landed before the write window opens (an FFT pass computes here)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wait_write();
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const long long t = t0 + d;
            if (t < ntiles) {
#pragma unroll
                for (int k = 0; k < E; ++k) __builtin_nontemporal_store(v[d][k], b + t * TILE + k * THREADS + tid);
            }
        }
    }
    if (tid == 0 && spins) atomicAdd(spins, waited);
}

// free-running with prefetch of the next tile (what the FFT passes do): reference for mode 0
template <int THREADS, int E>
__global__ void __launch_bounds__(THREADS) prefetch_kernel(const d2v* __restrict__ a, d2v* __restrict__ b, long long ntiles) {
    const int     tid = threadIdx.x;
    constexpr int TILE = THREADS * E;
    d2v           v[E], vn[E];
    long long     t = blockIdx.x;
    if (t < ntiles) {
#pragma unroll
        for (int k = 0; k < E; ++k) v[k] = __builtin_nontemporal_load(a + t * TILE + k * THREADS + tid);
    }
    for (; t < ntiles; t += gridDim.x) {
        const long long tn = t + gridDim.x;
        if (tn < ntiles) {
#pragma unroll
            for (int k = 0; k < E; ++k) vn[k] = __builtin_nontemporal_load(a + tn * TILE + k * THREADS + tid);
        }
#pragma unroll
        for (int k = 0; k < E; ++k) __builtin_nontemporal_store(v[k], b + t * TILE + k * THREADS + tid);
#pragma unroll
        for (int k = 0; k < E; ++k) v[k] = vn[k];
    }
}

static hipStream_t g_s;
template <class F> static double time_ms(F&& launch, int reps = 7) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int r = 0; r < reps + 2; ++r) {
        CK(hipEventRecord(e0, g_s));
        launch();
        CK(hipEventRecord(e1, g_s));
        CK(hipEventSynchronize(e1));
        float m;
        CK(hipEventElapsedTime(&m, e0, e1));
        if (r >= 2) ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 2.0;
    const size_t bytes = (size_t)(gib * (1ull << 30));
    CK(hipStreamCreate(&g_s));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    d2v *a, *b;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes));
    CK(hipMemset(b, 0, bytes));
    unsigned long long* spins;
    CK(hipMalloc(&spins, 8));
    printf("# %.1f GiB -> %.1f GiB, %d CUs; rate = 2 x bytes / time\n", gib, gib, cus);
    constexpr int THREADS = 512, E = 8;
    const long long ntiles = (long long)(bytes / 16 / (THREADS * E));
    {
        const double ms = time_ms([&] { hipLaunchKernelGGL((prefetch_kernel<THREADS, E>), dim3(cus), dim3(THREADS), 0, g_s, a, b, ntiles); });
        printf("prefetch free-running, 1 WG/CU                      %.4f ms  %.0f GB/s\n", ms, 2.0 * bytes / ms * 1e-6);
        const double ms2 = time_ms([&] { hipLaunchKernelGGL((prefetch_kernel<THREADS, E>), dim3(2 * cus), dim3(THREADS), 0, g_s, a, b, ntiles); });
        printf("prefetch free-running, 2 WG/CU                      %.4f ms  %.0f GB/s\n", ms2, 2.0 * bytes / ms2 * 1e-6);
    }
    auto run = [&](auto kern, const char* name, int wgs_per_cu, int mode, unsigned period, unsigned rfrac) {
        CK(hipMemsetAsync(spins, 0, 8, g_s));
        const double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(cus * wgs_per_cu), dim3(THREADS), 0, g_s, a, b, ntiles, mode, period, rfrac, spins); });
        unsigned long long w = 0;
        CK(hipMemcpy(&w, spins, 8, hipMemcpyDeviceToHost));
        printf("%-10s wg/cu %d mode %d period %5.1f us read %3.0f %%   %.4f ms  %.0f GB/s   (sleep polls per WG per launch %.0f)\n", name, wgs_per_cu, mode,
               period * 0.01, rfrac / 2.56, ms, 2.0 * bytes / ms * 1e-6, (double)w / 9.0 / (cus * wgs_per_cu));
    };
    for (int wg : {1, 2}) {
        run(tdm_kernel<THREADS, E, 1>, "depth1", wg, 0, 100, 128);
        run(tdm_kernel<THREADS, E, 2>, "depth2", wg, 0, 100, 128);
        run(tdm_kernel<THREADS, E, 4>, "depth4", wg, 0, 100, 128);
    }
    for (int mode : {1, 2})
        for (int wg : {1, 2})
            for (unsigned period : {300u, 500u, 800u, 1200u, 2000u, 4000u})
                for (unsigned rfrac : {112u, 128u, 144u}) {
                    run(tdm_kernel<THREADS, E, 1>, "depth1", wg, mode, period, rfrac);
                    if (period >= 800) run(tdm_kernel<THREADS, E, 2>, "depth2", wg, mode, period, rfrac);
                    if (period >= 1200) run(tdm_kernel<THREADS, E, 4>, "depth4", wg, mode, period, rfrac);
                }
    // correctness of the last copy
    std::vector<unsigned char> h(4096);
    CK(hipMemcpy(h.data(), (char*)b + bytes - 4096, 4096, hipMemcpyDeviceToHost));
    int bad = 0;
    for (unsigned char c : h) bad += c != 1;
    printf("# check: %d bad bytes in the last page\n", bad);
    return 0;
}
