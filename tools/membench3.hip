// tools/membench3.hip -- developer measurement: copy 2 GiB -> 2 GiB with FEW fat waves.  membench2 showed that HBM writes
// run at 6.5 TB/s from 1-4 waves per CU and at 4.5 TB/s from 16, while reads want many requests in flight; here every lane
// keeps U 16-byte loads in flight (optionally the next chunk's loads are issued before the current chunk's stores), a
// workgroup moves whole contiguous chunks of THREADS*U*16 bytes, and chunks are handed out grid-strided or as one contiguous
// region per workgroup.
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

// chunk = THREADS * U elements, lane-interleaved inside (element = chunk*THREADS*U + u*THREADS + tid)
template <int THREADS, int U, bool PF, bool CONTIG, bool NTL, bool NTS>
__global__ void __launch_bounds__(THREADS) fat_copy(const d2* in, d2* out, size_t nchunks) {
    const size_t per = (nchunks + gridDim.x - 1) / gridDim.x;
    size_t       c = CONTIG ? per * blockIdx.x : blockIdx.x;
    const size_t cend = CONTIG ? (per * (blockIdx.x + 1) < nchunks ? per * (blockIdx.x + 1) : nchunks) : nchunks;
    const size_t step = CONTIG ? 1 : gridDim.x;
    d2 v[U], w[PF ? U : 1];
    auto load = [&](size_t ch, d2* d) {
#pragma unroll
        for (int u = 0; u < U; ++u) d[u] = ld<NTL>(in + ch * (THREADS * U) + u * THREADS + threadIdx.x);
    };
    if (PF && c < cend) load(c, v);
    for (; c < cend; c += step) {
        if (PF) {
            if (c + step < cend) load(c + step, w);
        } else {
            load(c, v);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) st<NTS>(out + c * (THREADS * U) + u * THREADS + threadIdx.x, v[u]);
        if (PF) {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = w[u];
        }
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

static d2 *g_a, *g_b;
static const size_t N = 512ull * 512 * 512;

template <int THREADS, int U, bool PF, bool CONTIG, bool NTL, bool NTS> void run(int grid) {
    const size_t nchunks = N / (THREADS * U);
    double ms = time_ms([&] { hipLaunchKernelGGL((fat_copy<THREADS, U, PF, CONTIG, NTL, NTS>), dim3(grid), dim3(THREADS), 0, g_s, g_a, g_b, nchunks); });
    printf("copy %4d thr U%-2d %s %s %s%s grid %5d  %7.3f ms  %6.0f GB/s\n", THREADS, U, PF ? "pf  " : "nopf", CONTIG ? "block-contig" : "grid-stride ",
           NTL ? "ntl" : "   ", NTS ? "nts" : "   ", grid, ms, 2.0 * N * 16 / ms * 1e-6);
    fflush(stdout);
}
template <int THREADS, int U> void sweep() {
    for (int grid : {128, 256, 512, 1024}) {
        if (grid * THREADS > 256 * 2048) continue;
        run<THREADS, U, false, false, true, true>(grid);
        run<THREADS, U, true, false, true, true>(grid);
        run<THREADS, U, false, true, true, true>(grid);
        run<THREADS, U, true, true, true, true>(grid);
        run<THREADS, U, true, false, false, false>(grid);
        run<THREADS, U, true, false, true, false>(grid);
    }
}

int main() {
    CK(hipMalloc(&g_a, N * 16));
    CK(hipMalloc(&g_b, N * 16));
    CK(hipMemset(g_a, 1, N * 16));
    CK(hipMemset(g_b, 0, N * 16));
    CK(hipStreamCreateWithFlags(&g_s, hipStreamNonBlocking));
    sweep<256, 4>();
    sweep<256, 8>();
    sweep<256, 16>();
    sweep<512, 4>();
    sweep<512, 8>();
    sweep<512, 16>();
    sweep<1024, 4>();
    sweep<1024, 8>();
    sweep<64, 8>();
    sweep<64, 16>();
    sweep<128, 16>();
    return 0;
}
