// tools/zy_stream.hip -- developer experiment: t0 (2D YZ FFT of every plane) as ONE persistent launch in which Z rows and Y columns
// of planes LAG apart are in flight at the same time (tools/experimental/dfft_zy_stream.h), against the library's two-launch,
// cache-chunked t0.  Every variant is verified element by element against the library's row + column kernels before it is timed.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I distributedfft_amd/csrc -I include -I tools tools/zy_stream.hip
//          -L distributedfft_amd/lib -ldfft_mi355x -Wl,-rpath,$PWD/distributedfft_amd/lib -o tools/bin/zy_stream
//   usage: zy_stream [rounds] [variant substring]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "dfft.h"
#include "experimental/dfft_zy_stream.h"

using namespace dfft;

#define CK(...)                                                                           \
    do {                                                                                  \
        hipError_t e_ = (__VA_ARGS__);                                                    \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

__global__ void fill_kernel(double2* a, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long h = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        h ^= h >> 29;
        h *= 0xBF58476D1CE4E5B9ull;
        h ^= h >> 32;
        a[i] = double2{(double)(h & 0xFFFFF) / 1048576.0 - 0.5, (double)((h >> 20) & 0xFFFFF) / 1048576.0 - 0.5};
    }
}
// a: [plane][n1*n2] with planes a_plane elements apart, b: natural
__global__ void diff_kernel(const double2* a, long long a_plane, const double2* b, long long plane, size_t n, double* maxdiff, double* maxref) {
    double d = 0, r = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t   p = i / plane, o = i - p * plane;
        const double2  x = a[p * a_plane + o], y = b[i];
        d = fmax(d, fmax(fabs(x.x - y.x), fabs(x.y - y.y)));
        r = fmax(r, fmax(fabs(y.x), fabs(y.y)));
    }
    atomicMax((unsigned long long*)maxdiff, (unsigned long long)__double_as_longlong(d));  // doubles >= 0 compare like their bits
    atomicMax((unsigned long long*)maxref, (unsigned long long)__double_as_longlong(r));
}

static double2* make_tw(int n) {
    std::vector<double> h(2 * (size_t)n);
    const long double   two_pi = 6.283185307179586476925286766559005768L;
    for (int k = 0; k < n; ++k) {
        const long double a = two_pi * (long double)k / (long double)n;
        h[2 * k] = (double)cosl(a);
        h[2 * k + 1] = (double)(-sinl(a));
    }
    double2* d;
    CK(hipMalloc(&d, h.size() * sizeof(double)));
    CK(hipMemcpy(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    return d;
}

using P512 = Plan<512, 8, 8, 8, 8>;

struct CfgFence : ZyCfgDefault { static constexpr int HANDOFF = 1; };
struct CfgNoPF : ZyCfgDefault { static constexpr bool PREFETCH = false; };
struct CfgCoarse : ZyCfgDefault { static constexpr bool FINE = false; };
struct CfgOutNT : ZyCfgDefault { static constexpr bool OUT_NT = true; };
struct CfgStatic : ZyCfgDefault { static constexpr bool DYNAMIC = false; };
struct CfgStaticFence : ZyCfgDefault { static constexpr bool DYNAMIC = false; static constexpr int HANDOFF = 1; };
struct CfgB8 : ZyCfgDefault { static constexpr int BUNDLE = 8; };
struct CfgB8Fence : ZyCfgDefault { static constexpr int BUNDLE = 8; static constexpr int HANDOFF = 1; };
struct CfgB16Fence : ZyCfgDefault { static constexpr int BUNDLE = 16; static constexpr int HANDOFF = 1; };
struct CfgChunk64 : ZyCfgDefault { static constexpr int CHUNK = 64; };
struct CfgChunk64B8Fence : ZyCfgDefault { static constexpr int CHUNK = 64; static constexpr int BUNDLE = 8; static constexpr int HANDOFF = 1; };
struct CfgChunk64B16Fence : ZyCfgDefault { static constexpr int CHUNK = 64; static constexpr int BUNDLE = 16; static constexpr int HANDOFF = 1; };
struct CfgChunk32B8Fence : ZyCfgDefault { static constexpr int CHUNK = 32; static constexpr int BUNDLE = 8; static constexpr int HANDOFF = 1; };
struct CfgChunk64B8FenceStatic : CfgChunk64B8Fence { static constexpr bool DYNAMIC = false; };
struct CfgSkelChunk64B8Fence : CfgChunk64B8Fence { static constexpr bool MATH = false; };
struct CfgSkel : ZyCfgDefault { static constexpr bool MATH = false; };
struct CfgSkelFence : ZyCfgDefault { static constexpr bool MATH = false; static constexpr int HANDOFF = 1; };

struct Ctx {
    double2 *in, *w, *ref, *tw;
    ZyCtl*   ctl;
    double * dmax, *rmax;
    size_t   n;
    long long w_plane;
    int      cus, rounds;
    hipStream_t s;
};

static const char* g_filter = nullptr;
template <class Cfg> void run_variant(Ctx& c, const char* name, unsigned lag, int bpc_req) {
    char full[128];
    snprintf(full, sizeof(full), "%s lag %u x%d/CU", name, lag, bpc_req);
    if (g_filter && !strstr(full, g_filter)) return;
    using G = ZyGeom<double2, P512, P512, Cfg>;
    auto kern = zy_stream_kernel<double2, P512, P512, +1, Cfg>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, Cfg::THREADS, G::LDS_BYTES));
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern)));
    const int bpc = std::min(occ, bpc_req);
    if (bpc < 1) {
        printf("%-44s occupancy 0, skipped\n", full);
        return;
    }
    const unsigned grid = (unsigned)(c.cus * bpc);
    auto launch = [&]() {
        CK(hipMemsetAsync(c.ctl, 0, sizeof(ZyCtl), c.s));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::THREADS), G::LDS_BYTES, c.s, (const double2*)c.in, c.w, c.ctl, (const double2*)c.tw,
                           (const double2*)c.tw, 512ll * 512, c.w_plane, 512u, lag);
        CK(hipGetLastError());
    };
    CK(hipMemsetAsync(c.w, 0xff, (size_t)512 * c.w_plane * 16, c.s));
    launch();
    CK(hipStreamSynchronize(c.s));
    ZyCtl* h = new ZyCtl;
    CK(hipMemcpy(h, c.ctl, sizeof(ZyCtl), hipMemcpyDeviceToHost));
    if (h->error) {
        printf("%-44s grid %u (occ %d, vgpr %d, scratch %zu, lds %zu) ERROR %u  ticket %u\n", full, grid, occ, fa.numRegs, (size_t)fa.localSizeBytes,
               G::LDS_BYTES, h->error, h->ticket);
        delete h;
        return;
    }
    CK(hipMemset(c.dmax, 0, 8));
    CK(hipMemset(c.rmax, 0, 8));
    hipLaunchKernelGGL(diff_kernel, dim3(2048), dim3(256), 0, c.s, (const double2*)c.w, c.w_plane, (const double2*)(Cfg::MATH ? c.ref : c.in),
                       512ll * 512, c.n, c.dmax, c.rmax);
    CK(hipStreamSynchronize(c.s));
    double dm, rm;
    CK(hipMemcpy(&dm, c.dmax, 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&rm, c.rmax, 8, hipMemcpyDeviceToHost));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> ms;
    unsigned           errs = 0, waits = 0;
    for (int r = 0; r < c.rounds; ++r) {
        CK(hipEventRecord(e0, c.s));
        launch();
        CK(hipEventRecord(e1, c.s));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t);
        CK(hipMemcpy(h, c.ctl, 3 * 128, hipMemcpyDeviceToHost));
        errs += h->error != 0;
        waits = h->waits;
    }
    std::sort(ms.begin(), ms.end());
    printf("%-44s grid %4u (occ %d, vgpr %3d, scratch %3zu, lds %6zu) median %.3f ms  min %.3f ms  rel.err %.2e  errs %u  waits %u\n", full, grid,
           occ, fa.numRegs, (size_t)fa.localSizeBytes, G::LDS_BYTES, ms[ms.size() / 2], ms[0], dm / rm, errs, waits);
    fflush(stdout);
    delete h;
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
    Ctx c;
    c.rounds = argc > 1 ? atoi(argv[1]) : 9;
    if (argc > 2) g_filter = argv[2];
    c.n = 512ull * 512 * 512;
    c.w_plane = 512ll * 512 + 24;  // the plan's padded work buffer: planes 3 cache lines further apart
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    c.cus = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, c.cus);
    CK(hipMalloc(&c.in, c.n * 16));
    CK(hipMalloc(&c.w, (size_t)512 * c.w_plane * 16));
    CK(hipMalloc(&c.ref, c.n * 16));
    CK(hipMalloc(&c.ctl, sizeof(ZyCtl)));
    CK(hipMalloc(&c.dmax, 8));
    CK(hipMalloc(&c.rmax, 8));
    c.tw = make_tw(512);
    CK(hipStreamCreateWithFlags(&c.s, hipStreamNonBlocking));
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, c.s, c.in, c.n, 12345u);
    CK(hipStreamSynchronize(c.s));
    // reference: the library's row and column kernels over the whole slab
    if (dfft_fft1d_rows(c.in, c.ref, 512, 512ll * 512, DFFT_F64, DFFT_FORWARD, c.s) ||
        dfft_fft1d_cols(c.ref, c.ref, 512, 512, 512, DFFT_F64, DFFT_FORWARD, c.s)) {
        printf("library reference failed: %s\n", dfft_last_error());
        return 1;
    }
    CK(hipStreamSynchronize(c.s));
    // baseline: t0 of the library's 512^3 plan (two launches per cache chunk)
    {
        double2* out;
        CK(hipMalloc(&out, c.n * 16));
        dfft_plan_t plan;
        if (dfft_plan_create(&plan, 512, 512, 512, DFFT_F64, DFFT_FORWARD, c.in, out, nullptr, 0, 1, DFFT_PLAN_INPUT_FROM_IN)) {
            printf("plan failed: %s\n", dfft_last_error());
            return 1;
        }
        std::vector<double> t0s, tot;
        for (int r = 0; r < c.rounds + 10; ++r) {
            double t[4];
            dfft_execute(plan, DFFT_EXEC_ASYNC);
            dfft_stage_times(plan, t);
            if (r >= 10) {
                t0s.push_back(t[0] * 1e3);
                tot.push_back((t[0] + t[1] + t[2] + t[3]) * 1e3);
            }
        }
        std::sort(t0s.begin(), t0s.end());
        std::sort(tot.begin(), tot.end());
        printf("%-44s t0 median %.3f ms  min %.3f ms   (whole transform median %.3f ms)\n", "library: chunked two-launch t0", t0s[t0s.size() / 2],
               t0s[0], tot[tot.size() / 2]);
        dfft_plan_destroy(plan);
        CK(hipFree(out));
    }
    // phase order (the library's 64-plane cache chunks inside one launch) with the hand-off cost paid once per bundle of units
    run_variant<CfgChunk64B8Fence>(c, "chunk 64, bundle 8, plain + fences", 0, 1);
    run_variant<CfgChunk64B16Fence>(c, "chunk 64, bundle 16, plain + fences", 0, 1);
    run_variant<CfgChunk32B8Fence>(c, "chunk 32, bundle 8, plain + fences", 0, 1);
    run_variant<CfgChunk64B8FenceStatic>(c, "chunk 64, bundle 8, fences, static", 0, 1);
    run_variant<CfgSkelChunk64B8Fence>(c, "chunk 64, bundle 8, fences, no math", 0, 1);
    run_variant<CfgChunk64>(c, "chunk 64, sc1 hand-off", 0, 1);
    run_variant<CfgB8Fence>(c, "bundle 8, plain + fences", 32, 1);
    run_variant<CfgB16Fence>(c, "bundle 16, plain + fences", 32, 1);
    run_variant<CfgB8>(c, "bundle 8, sc1 hand-off", 32, 1);
    for (unsigned lag : {64u, 16u, 8u, 32u, 4u}) {
        run_variant<ZyCfgDefault>(c, "stream sc1 hand-off", lag, 1);
        run_variant<CfgFence>(c, "stream plain + fences", lag, 1);
    }
    run_variant<CfgStatic>(c, "stream sc1, static ticket rotation", 16, 1);
    run_variant<CfgStaticFence>(c, "stream fences, static ticket rotation", 16, 1);
    run_variant<CfgNoPF>(c, "stream sc1, no prefetch", 16, 1);
    run_variant<CfgCoarse>(c, "stream sc1, coarse ticket order", 16, 1);
    run_variant<CfgOutNT>(c, "stream sc1, nt result stores", 16, 1);
    run_variant<CfgSkel>(c, "skeleton (no math) sc1", 16, 1);
    run_variant<CfgSkelFence>(c, "skeleton (no math) fences", 16, 1);
    run_variant<ZyCfgDefault>(c, "stream sc1 hand-off", 16, 2);
    return 0;
}
