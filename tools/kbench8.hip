// tools/kbench8.hip -- developer experiment: the library's own variant selection (launch_plan) on long column FFTs
// (N = 768, 1024, 2048; fp64 and fp32 column pairs; Y-like column->column and X-like transposed-store launches).
// Built several times with different -DDFFT_TW_EFFECTIVE / -DDFFT_PREFETCH_MAX_REGS to A/B the selection rules.
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
#define CK(...)                                                                           \
    do {                                                                                  \
        hipError_t e_ = (__VA_ARGS__);                                                    \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

static AxisMap plain_axis(long long n, long long stride, long long cstride) { return AxisMap{(int)n, 1, 0, stride, cstride, 0}; }

template <class W> W* make_tw(int n) {
    W* tw;
    CK(hipMalloc(&tw, n * sizeof(W)));
    std::vector<W> h(n);
    for (int k = 0; k < n; ++k) {
        h[k].x = cos(2 * M_PI * k / n);
        h[k].y = -sin(2 * M_PI * k / n);
    }
    CK(hipMemcpy(tw, h.data(), n * sizeof(W), hipMemcpyHostToDevice));
    return tw;
}

using P768 = Plan<768, 24, 8, 8, 4, 3>;
using P1024 = Plan<1024, 16, 8, 8, 8, 2>;
using P2048 = Plan<2048, 16, 8, 8, 8, 4>;

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 7;
    const long long bytes = 1ll << 31;  // 2 GiB per buffer
    void *a, *b;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    {
        std::vector<float> x(1 << 20);
        for (auto& v : x) v = ((float)rand() / RAND_MAX * 2 - 1) * 1e-3f;
        for (long long off = 0; off < bytes; off += (4 << 20)) CK(hipMemcpy((char*)a + off, x.data(), 4 << 20, hipMemcpyHostToDevice));
    }
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    // slab [n][ys][n2] -> X-like: out [ys][n2][n];  Y-like: planes [na][n][n2] in place layout to b
    auto mk = [&](int dtype, int n, const void* tw, bool xlike) {
        const long long S = dtype == F64 ? 16 : 8;
        const int       n2 = 512;
        const long long ys = bytes / S / n / n2;
        FftLaunch L;
        memset(&L, 0, sizeof(L));
        L.dtype = dtype; L.n = n; L.dir = 1; L.cols = 1; L.in = a; L.out = b; L.tw = tw;
        if (xlike) {
            L.imap = plain_axis(n, ys * n2, 1);
            L.itile = TileMap{n2, 1};
            L.omap = plain_axis(n, 1, n);
            L.otile = TileMap{(long long)n2 * n, (long long)n};
        } else {
            L.imap = L.omap = plain_axis(n, n2, 1);
            L.itile = L.otile = TileMap{(long long)n * n2, 1};
        }
        L.na = ys; L.ncols = n2;
        return L;
    };
    struct Case { std::string name; std::function<hipError_t()> run; };
    std::vector<Case> cases;
    double2* t64[3] = {make_tw<double2>(768), make_tw<double2>(1024), make_tw<double2>(2048)};
    float2*  t32[3] = {make_tw<float2>(768), make_tw<float2>(1024), make_tw<float2>(2048)};
    auto add = [&](const char* nm, auto plan, int n, int ti) {
        using P = decltype(plan);
        for (int xl = 0; xl < 2; ++xl) {
            FftLaunch L64 = mk(F64, n, t64[ti], xl), L32 = mk(F32, n, t32[ti], xl);
            cases.push_back({std::string(nm) + (xl ? " X-like" : " Y-like") + " fp64", [=] { return launch_plan<double2, P>(L64, s); }});
            cases.push_back({std::string(nm) + (xl ? " X-like" : " Y-like") + " pair", [=] {
                                 FftLaunch Lp;
                                 if (!make_pair_launch<P>(L32, Lp)) return hipErrorInvalidValue;
                                 return launch_plan<cpair, P>(Lp, s);
                             }});
        }
    };
    add("768 ", P768{}, 768, 0);
    add("1024", P1024{}, 1024, 1);
    add("2048", P2048{}, 2048, 2);

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
    printf("TW_EFFECTIVE=%d PREFETCH_MAX_REGS=%d\n", (int)DFFT_TW_EFFECTIVE, (int)DFFT_PREFETCH_MAX_REGS);
    for (size_t i = 0; i < cases.size(); ++i) {
        auto v = ms[i];
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        printf("%-28s median %.3f ms  %.0f GB/s\n", cases[i].name.c_str(), v[v.size() / 2], 2.0 * bytes / v[v.size() / 2] / 1e6);
    }
    return 0;
}
