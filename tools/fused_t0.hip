// tools/fused_t0.hip -- developer experiment: t0 (2D YZ FFT of every plane) as ONE persistent kernel whose Z->Y
// intermediate stays in the XCD's L2 (tools/experimental/dfft_fused_yz.h), against the library's two-launch, cache-chunked t0.
// Every variant is verified element by element against the library's row + column kernels before it is timed.
//   usage: fused_t0 [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "dfft.h"
#include "experimental/dfft_fused_yz.h"

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
__global__ void diff_kernel(const double2* a, const double2* b, size_t n, double* maxdiff, double* maxref) {
    double d = 0, r = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        d = fmax(d, fmax(fabs(a[i].x - b[i].x), fabs(a[i].y - b[i].y)));
        r = fmax(r, fmax(fabs(b[i].x), fabs(b[i].y)));
    }
    // doubles >= 0 compare like their bit patterns
    atomicMax((unsigned long long*)maxdiff, (unsigned long long)__double_as_longlong(d));
    atomicMax((unsigned long long*)maxref, (unsigned long long)__double_as_longlong(r));
}

static double2* make_tw(int n) {
    std::vector<double> h(2 * (size_t)n);
    const long double two_pi = 6.283185307179586476925286766559005768L;
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

struct CfgBuf : FusedCfgDefault { static constexpr int SLOAD = 1; };
struct CfgT2 : FusedCfgDefault { static constexpr int TEAMS = 2; };
struct CfgT2Buf : FusedCfgDefault { static constexpr int TEAMS = 2; static constexpr int SLOAD = 1; };
struct CfgSkel : FusedCfgDefault { static constexpr bool MATH = false; };
struct CfgSkelT2 : FusedCfgDefault { static constexpr bool MATH = false; static constexpr int TEAMS = 2; };
struct CfgSNT : FusedCfgDefault { static constexpr bool SSTORE_NT = true; };
struct CfgNoPad : FusedCfgDefault { static constexpr int SPAD = 0; };
struct CfgNoPF : FusedCfgDefault { static constexpr bool PREFETCH = false; };
struct CfgW4 : FusedCfgDefault { static constexpr int MIN_WAVES = 4; };
struct CfgTR : FusedCfgDefault { static constexpr bool TW_RELOAD = true; };
struct CfgTRW4 : FusedCfgDefault { static constexpr bool TW_RELOAD = true; static constexpr int MIN_WAVES = 4; };
struct CfgTRW4NoPF : FusedCfgDefault { static constexpr bool TW_RELOAD = true; static constexpr int MIN_WAVES = 4; static constexpr bool PREFETCH = false; };
struct CfgShare : FusedCfgDefault { static constexpr int TEAMS = 2; static constexpr bool SHARE_S = true; };
struct CfgShareNoPF : CfgShare { static constexpr bool PREFETCH = false; };
struct CfgShareTR : CfgShare { static constexpr bool TW_RELOAD = true; };
struct CfgShareBuf : CfgShare { static constexpr int SLOAD = 1; };
struct CfgSkelShare : CfgShare { static constexpr bool MATH = false; };
struct CfgSkelShareNoPF : CfgShare { static constexpr bool MATH = false; static constexpr bool PREFETCH = false; };
struct CfgSkelSharePlain : CfgShare { static constexpr bool MATH = false; static constexpr bool IN_NT = false, OUT_NT = false; };
struct CfgSkelNoPF : FusedCfgDefault { static constexpr bool MATH = false; static constexpr bool PREFETCH = false; };
struct CfgW4T2 : FusedCfgDefault { static constexpr int MIN_WAVES = 4; static constexpr int TEAMS = 2; };

struct Ctx {
    double2 *in, *out, *ref, *scratch, *tw, *tw256;
    FusedCtl* ctl;
    double *  dmax, *rmax;
    size_t    n;
    int       cus, rounds;
    hipStream_t s;
};

static const char* g_filter = nullptr;
template <class Cfg> void run_variant(Ctx& c, const char* name, int bpc_req) {
    if (g_filter && !strstr(name, g_filter)) return;
    using FG = FusedGeom<double2, P512, P512, Cfg>;
    auto kern = fused_yz_kernel<double2, P512, P512, +1, Cfg>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FG::LDS_BYTES));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, Cfg::THREADS, FG::LDS_BYTES));
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern)));
    const int bpc = std::min(occ, bpc_req);
    if (bpc < 1) {
        printf("%-34s occupancy 0, skipped\n", name);
        return;
    }
    const unsigned grid = (unsigned)(c.cus * bpc);
    AxisMap omap;
    omap.blk = 512;
    omap.nblk = 1;
    omap.blk_stride = 0;
    omap.stride = 512;
    omap.cstride = 1;
    omap.last_delta = 0;
    omap.sub = 1;
    omap.sub_stride = 0;
    TileMap otile{512ll * 512, 1};
    auto launch = [&]() {
        CK(hipMemsetAsync(c.ctl, 0, sizeof(FusedCtl), c.s));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::THREADS), FG::LDS_BYTES, c.s, (const double2*)c.in, c.out, c.scratch, c.ctl,
                           (const double2*)c.tw, (const double2*)c.tw, omap, otile, 512ll * 512, 512u, 0u);
        CK(hipGetLastError());
    };
    CK(hipMemsetAsync(c.out, 0xff, c.n * 16, c.s));
    launch();
    CK(hipStreamSynchronize(c.s));
    FusedCtl h;
    CK(hipMemcpy(&h, c.ctl, sizeof(h), hipMemcpyDeviceToHost));
    std::string teams;
    for (int x = 0; x < 16; ++x)
        if (h.xcc_count[x]) teams += std::to_string(h.xcc_count[x]) + " ";
    if (h.error) {
        printf("%-34s grid %u (occ %d, vgpr %d, lds %zu) ERROR %u  registered %u  per-xcc: %s\n", name, grid, occ, fa.numRegs,
               FG::LDS_BYTES, h.error, h.registered, teams.c_str());
        return;
    }
    CK(hipMemset(c.dmax, 0, 8));
    CK(hipMemset(c.rmax, 0, 8));
    hipLaunchKernelGGL(diff_kernel, dim3(2048), dim3(256), 0, c.s, (const double2*)c.out, (const double2*)(Cfg::MATH ? c.ref : c.in), c.n,
                       c.dmax, c.rmax);
    CK(hipStreamSynchronize(c.s));
    double dm, rm;
    CK(hipMemcpy(&dm, c.dmax, 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&rm, c.rmax, 8, hipMemcpyDeviceToHost));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> ms;
    unsigned errs = 0;
    for (int r = 0; r < c.rounds; ++r) {
        CK(hipEventRecord(e0, c.s));
        launch();
        CK(hipEventRecord(e1, c.s));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t);
        CK(hipMemcpy(&h, c.ctl, sizeof(h), hipMemcpyDeviceToHost));
        errs += h.error != 0;
    }
    std::sort(ms.begin(), ms.end());
    printf("%-34s grid %4u (occ %d, vgpr %3d, lds %6zu) median %.3f ms  min %.3f ms  rel.err %.2e  errs %u  per-xcc: %s\n", name,
           grid, occ, fa.numRegs, FG::LDS_BYTES, ms[ms.size() / 2], ms[0], dm / rm, errs, teams.c_str());
    fflush(stdout);
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
}

using P256 = Plan<256, 8, 8, 8, 4>;
struct CfgSplit : FusedCfgDefault { static constexpr bool PREFETCH = false; };
struct CfgSplitPF : FusedCfgDefault { static constexpr bool PREFETCH = true; };
struct CfgSplitTR : FusedCfgDefault { static constexpr bool PREFETCH = false; static constexpr bool TW_RELOAD = true; };
struct CfgSplitBuf : FusedCfgDefault { static constexpr bool PREFETCH = false; static constexpr int SLOAD = 1; };
struct CfgSplitSkel : FusedCfgDefault { static constexpr bool PREFETCH = false; static constexpr bool MATH = false; };
struct CfgSplitSkelPF : FusedCfgDefault { static constexpr bool PREFETCH = true; static constexpr bool MATH = false; };
struct CfgSplitSkelPlain : FusedCfgDefault { static constexpr bool PREFETCH = false; static constexpr bool MATH = false; static constexpr bool IN_NT = false, OUT_NT = false; };

template <class Cfg> void run_split(Ctx& c, const char* name) {
    if (g_filter && !strstr(name, g_filter)) return;
    using FG = FusedSplitGeom<double2, P512, P256, Cfg>;
    auto kern = fused_yz_split_kernel<double2, P512, P256, +1, Cfg>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FG::LDS_BYTES));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, Cfg::THREADS, FG::LDS_BYTES));
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern)));
    if (occ < 1) {
        printf("%-34s occupancy 0, skipped\n", name);
        return;
    }
    const unsigned grid = (unsigned)c.cus;
    AxisMap omap;
    omap.blk = 512;
    omap.nblk = 1;
    omap.blk_stride = 0;
    omap.stride = 512;
    omap.cstride = 1;
    omap.last_delta = 0;
    omap.sub = 1;
    omap.sub_stride = 0;
    TileMap otile{512ll * 512, 1};
    auto launch = [&]() {
        CK(hipMemsetAsync(c.ctl, 0, sizeof(FusedCtl), c.s));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::THREADS), FG::LDS_BYTES, c.s, (const double2*)c.in, c.out, c.scratch, c.ctl,
                           (const double2*)c.tw, (const double2*)c.tw256, (const double2*)c.tw, omap, otile, 512ll * 512, 512u, 0u);
        CK(hipGetLastError());
    };
    CK(hipMemsetAsync(c.out, 0xff, c.n * 16, c.s));
    launch();
    CK(hipStreamSynchronize(c.s));
    FusedCtl h;
    CK(hipMemcpy(&h, c.ctl, sizeof(h), hipMemcpyDeviceToHost));
    if (h.error) {
        printf("%-34s grid %u (occ %d, vgpr %d, lds %zu) ERROR %u  registered %u\n", name, grid, occ, fa.numRegs, FG::LDS_BYTES, h.error,
               h.registered);
        return;
    }
    CK(hipMemset(c.dmax, 0, 8));
    CK(hipMemset(c.rmax, 0, 8));
    hipLaunchKernelGGL(diff_kernel, dim3(2048), dim3(256), 0, c.s, (const double2*)c.out, (const double2*)(Cfg::MATH ? c.ref : c.in), c.n,
                       c.dmax, c.rmax);
    CK(hipStreamSynchronize(c.s));
    double dm, rm;
    CK(hipMemcpy(&dm, c.dmax, 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&rm, c.rmax, 8, hipMemcpyDeviceToHost));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> ms;
    unsigned errs = 0;
    for (int r = 0; r < c.rounds; ++r) {
        CK(hipEventRecord(e0, c.s));
        launch();
        CK(hipEventRecord(e1, c.s));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t);
        CK(hipMemcpy(&h, c.ctl, sizeof(h), hipMemcpyDeviceToHost));
        errs += h.error != 0;
    }
    std::sort(ms.begin(), ms.end());
    printf("%-34s grid %4u (occ %d, vgpr %3d, lds %6zu) median %.3f ms  min %.3f ms  rel.err %.2e  errs %u\n", name, grid, occ, fa.numRegs,
           FG::LDS_BYTES, ms[ms.size() / 2], ms[0], dm / rm, errs);
    fflush(stdout);
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
    Ctx c;
    c.rounds = argc > 1 ? atoi(argv[1]) : 9;
    if (argc > 2) g_filter = argv[2];
    c.n = 512ull * 512 * 512;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    c.cus = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, c.cus);
    CK(hipMalloc(&c.in, c.n * 16));
    CK(hipMalloc(&c.out, c.n * 16));
    CK(hipMalloc(&c.ref, c.n * 16));
    CK(hipMalloc(&c.scratch, 32 * 512ull * (512 + 8) * 16));
    CK(hipMalloc(&c.ctl, sizeof(FusedCtl)));
    CK(hipMalloc(&c.dmax, 8));
    CK(hipMalloc(&c.rmax, 8));
    c.tw = make_tw(512);
    c.tw256 = make_tw(256);
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
        dfft_plan_t plan;
        if (dfft_plan_create(&plan, 512, 512, 512, DFFT_F64, DFFT_FORWARD, c.in, c.out, nullptr, 0, 1, DFFT_PLAN_INPUT_FROM_IN)) {
            printf("plan failed: %s\n", dfft_last_error());
            return 1;
        }
        std::vector<double> t0s, tot;
        for (int r = 0; r < c.rounds + 2; ++r) {
            double t[4];
            dfft_execute(plan, DFFT_EXEC_ASYNC);
            dfft_stage_times(plan, t);
            if (r >= 2) {
                t0s.push_back(t[0] * 1e3);
                tot.push_back((t[0] + t[1] + t[2] + t[3]) * 1e3);
            }
        }
        std::sort(t0s.begin(), t0s.end());
        std::sort(tot.begin(), tot.end());
        printf("%-34s t0 median %.3f ms  min %.3f ms   (whole transform median %.3f ms)\n", "library: chunked two-launch t0", t0s[t0s.size() / 2],
               t0s[0], tot[tot.size() / 2]);
        dfft_plan_destroy(plan);
    }
    run_split<CfgSplit>(c, "split fused");
    run_split<CfgSplitPF>(c, "split fused prefetch");
    run_split<CfgSplitTR>(c, "split fused tw reload");
    run_split<CfgSplitBuf>(c, "split fused sc1-bufload");
    run_split<CfgSplitSkel>(c, "split skeleton");
    run_split<CfgSplitSkelPF>(c, "split skeleton prefetch");
    run_split<CfgSplitSkelPlain>(c, "split skeleton plain in/out");
    for (int bpc : {2, 1}) {
        const std::string sfx = bpc == 2 ? " x2/CU" : " x1/CU";
        run_variant<FusedCfgDefault>(c, ("fused nt-load" + sfx).c_str(), bpc);
        run_variant<CfgBuf>(c, ("fused sc1-bufload" + sfx).c_str(), bpc);
        run_variant<CfgT2>(c, ("fused 2 teams/XCD" + sfx).c_str(), bpc);
        run_variant<CfgT2Buf>(c, ("fused 2 teams sc1-bufload" + sfx).c_str(), bpc);
        run_variant<CfgSNT>(c, ("fused S stores nt" + sfx).c_str(), bpc);
        run_variant<CfgNoPad>(c, ("fused no S padding" + sfx).c_str(), bpc);
        run_variant<CfgNoPF>(c, ("fused no prefetch" + sfx).c_str(), bpc);
        run_variant<CfgW4>(c, ("fused <=128 VGPR" + sfx).c_str(), bpc);
        run_variant<CfgW4T2>(c, ("fused <=128 VGPR 2 teams" + sfx).c_str(), bpc);
        run_variant<CfgTR>(c, ("fused tw reload" + sfx).c_str(), bpc);
        run_variant<CfgTRW4>(c, ("fused tw reload <=128" + sfx).c_str(), bpc);
        run_variant<CfgTRW4NoPF>(c, ("fused tw reload <=128 no pf" + sfx).c_str(), bpc);
        run_variant<CfgShare>(c, ("fused shared-S 2 teams" + sfx).c_str(), bpc);
        run_variant<CfgShareNoPF>(c, ("fused shared-S no prefetch" + sfx).c_str(), bpc);
        run_variant<CfgShareTR>(c, ("fused shared-S tw reload" + sfx).c_str(), bpc);
        run_variant<CfgShareBuf>(c, ("fused shared-S sc1-bufload" + sfx).c_str(), bpc);
        run_variant<CfgSkelShare>(c, ("skeleton shared-S" + sfx).c_str(), bpc);
        run_variant<CfgSkelShareNoPF>(c, ("skeleton shared-S no prefetch" + sfx).c_str(), bpc);
        run_variant<CfgSkelSharePlain>(c, ("skeleton shared-S plain in/out" + sfx).c_str(), bpc);
        run_variant<CfgSkelNoPF>(c, ("skeleton no prefetch" + sfx).c_str(), bpc);
        run_variant<CfgSkel>(c, ("skeleton (no math)" + sfx).c_str(), bpc);
        run_variant<CfgSkelT2>(c, ("skeleton 2 teams" + sfx).c_str(), bpc);
    }
    return 0;
}
