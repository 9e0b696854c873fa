// tools/kbench5.hip -- developer experiment: column-kernel geometries for N = 1024 / 2048 (BASELINE configs 4 and 5).
// X pass of a [N0][ys][512] fp64 slab -> [ys][512][N0], as one rank of config 4 (N0 = 1024, ys = 96) sees it.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "dfft_fft_impl.h"

using namespace dfft;
#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

template <bool OSTAGE_, bool NT_, int MINW_, bool PF_> struct Tune {
    static constexpr bool TWPOW = true;
    static constexpr bool OSTAGE = OSTAGE_;
    static constexpr bool NTL = NT_;
    static constexpr bool NTS = NT_;
    static constexpr int MIN_WAVES = MINW_;
    static constexpr int CB_OVERRIDE = 0;
    static constexpr bool PLAIN = false;
    static constexpr bool PREFETCH = PF_;
};
using P1024a = Plan<1024, 8, 8, 8, 8, 2>;    // T = 128
using P1024b = Plan<1024, 16, 8, 8, 8, 2>;   // T = 64
using P1024c = Plan<1024, 16, 8, 8, 4, 4>;   // T = 64, shorter last stages
using P2048a = Plan<2048, 16, 8, 8, 8, 4>;   // T = 128
using P2048b = Plan<2048, 32, 8, 8, 8, 4>;   // T = 64

static AxisMap plain_axis(long long n, long long stride, long long cstride) { return AxisMap{(int)n, 1, 0, stride, cstride, 0}; }

template <class V> V* make_tw(int n) {
    V* tw;
    CK(hipMalloc(&tw, n * sizeof(V)));
    std::vector<V> h(n);
    for (int k = 0; k < n; ++k) {
        h[k].x = cos(2 * M_PI * k / n);
        h[k].y = -sin(2 * M_PI * k / n);
    }
    CK(hipMemcpy(tw, h.data(), n * sizeof(V), hipMemcpyHostToDevice));
    return tw;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 7;
    const long long maxel = 1024ll * 96 * 512 * 2;  // 1.5 GiB of double2
    double2 *a, *b;
    CK(hipMalloc(&a, maxel * 16));
    CK(hipMalloc(&b, maxel * 16));
    {
        std::vector<double> x(1 << 20);
        for (auto& v : x) v = ((double)rand() / RAND_MAX * 2 - 1) * 1e-3;
        for (long long off = 0; off < maxel * 2; off += (1 << 20)) CK(hipMemcpy((double*)a + off, x.data(), (1 << 20) * 8, hipMemcpyHostToDevice));
    }
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    double2* tw1024 = make_tw<double2>(1024);
    double2* tw2048 = make_tw<double2>(2048);

    auto xpass = [&](int n0, long long ys, int n2, int cb, const void* tw) {
        FftLaunch L;
        memset(&L, 0, sizeof(L));
        L.dtype = F64; L.n = n0; L.dir = 1; L.cols = 1; L.in = a; L.out = b; L.tw = tw;
        L.imap = plain_axis(n0, ys * n2, 1);
        L.itile = TileMap{n2, 1};
        L.omap = plain_axis(n0, 1, n0);
        L.otile = TileMap{(long long)n2 * n0, (long long)n0};
        L.tiles_per_a = n2 / cb; L.ntiles = ys * (n2 / cb); L.ncols = n2;
        return L;
    };
    struct Case { std::string name; std::function<hipError_t()> run; double bytes; };
    std::vector<Case> cases;
    const double b1024 = 2.0 * 16 * 1024 * 96 * 512, b2048 = 2.0 * 16 * 2048 * 64 * 512;
    FftLaunch X8 = xpass(1024, 96, 512, 8, tw1024), X4 = xpass(1024, 96, 512, 4, tw1024);
    FftLaunch Y8 = xpass(2048, 64, 512, 8, tw2048), Y4 = xpass(2048, 64, 512, 4, tw2048);
    //                                                                      OSTAGE NT  W  PF
    cases.push_back({"1024 E8  T128 cb8 1024thr direct (library now)", [&] { return launch_variant<double2, P1024a, 8, 1, 1, false, Tune<false, false, 0, false>>(X8, s); }, b1024});
    cases.push_back({"1024 E8  T128 cb8 staged nt", [&] { return launch_variant<double2, P1024a, 8, 1, 1, false, Tune<true, true, 0, false>>(X8, s); }, b1024});
    cases.push_back({"1024 E16 T64  cb8  512thr direct", [&] { return launch_variant<double2, P1024b, 8, 1, 1, false, Tune<false, false, 0, false>>(X8, s); }, b1024});
    cases.push_back({"1024 E16 T64  cb8  512thr staged nt", [&] { return launch_variant<double2, P1024b, 8, 1, 1, false, Tune<true, true, 0, false>>(X8, s); }, b1024});
    cases.push_back({"1024 E16 T64  cb8  {8,8,4,4} staged nt", [&] { return launch_variant<double2, P1024c, 8, 1, 1, false, Tune<true, true, 0, false>>(X8, s); }, b1024});
    cases.push_back({"1024 E16 T64  cb4  256thr staged nt (2 blk/CU)", [&] { return launch_variant<double2, P1024b, 4, 1, 1, false, Tune<true, true, 0, false>>(X4, s); }, b1024});
    cases.push_back({"1024 E8  T128 cb4  512thr staged nt prefetch", [&] { return launch_variant<double2, P1024a, 4, 1, 1, false, Tune<true, true, 0, true>>(X4, s); }, b1024});
    cases.push_back({"1024 E8  T128 cb4  512thr direct prefetch", [&] { return launch_variant<double2, P1024a, 4, 1, 1, false, Tune<false, false, 0, true>>(X4, s); }, b1024});
    cases.push_back({"2048 E16 T128 cb4  512thr direct (library now)", [&] { return launch_variant<double2, P2048a, 4, 1, 1, false, Tune<false, false, 0, false>>(Y4, s); }, b2048});
    cases.push_back({"2048 E16 T128 cb4  512thr staged nt", [&] { return launch_variant<double2, P2048a, 4, 1, 1, false, Tune<true, true, 0, false>>(Y4, s); }, b2048});
    cases.push_back({"2048 E32 T64  cb4  256thr staged nt", [&] { return launch_variant<double2, P2048b, 4, 1, 1, false, Tune<true, true, 0, false>>(Y4, s); }, b2048});
    cases.push_back({"2048 E32 T64  cb4  256thr direct", [&] { return launch_variant<double2, P2048b, 4, 1, 1, false, Tune<false, false, 0, false>>(Y4, s); }, b2048});

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<std::vector<float>> ms(cases.size());
    for (int r = 0; r < rounds + 1; ++r)
        for (size_t i = 0; i < cases.size(); ++i) {
            CK(hipEventRecord(e0, s));
            hipError_t e = cases[i].run();
            if (e != hipSuccess) { if (r == 0) printf("%s failed: %s\n", cases[i].name.c_str(), hipGetErrorString(e)); (void)hipGetLastError(); continue; }
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float t;
            CK(hipEventElapsedTime(&t, e0, e1));
            if (r > 0) ms[i].push_back(t);
        }
    for (size_t i = 0; i < cases.size(); ++i) {
        auto v = ms[i];
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        printf("%-52s median %.3f ms  %.0f GB/s\n", cases[i].name.c_str(), v[v.size() / 2], cases[i].bytes / v[v.size() / 2] / 1e6);
    }
    return 0;
}
