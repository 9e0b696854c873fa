// tools/kbench4.hip -- developer experiment: register prefetch of the next tile (Tune::PREFETCH) on the three passes, 512^3 fp64.
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
using P512 = Plan<512, 8, 8, 8, 8>;

static AxisMap plain_axis(long long n, long long stride, long long cstride) { return AxisMap{(int)n, 1, 0, stride, cstride, 0}; }

int main(int argc, char** argv) {
    const int n = 512, rounds = argc > 1 ? atoi(argv[1]) : 7;
    const long long N = (long long)n * n * n, nn = (long long)n * n;
    double2 *a, *b, *tw;
    CK(hipMalloc(&a, N * 16));
    CK(hipMalloc(&b, N * 16));
    CK(hipMalloc(&tw, n * 16));
    {
        std::vector<double> h(2 * (size_t)n);
        for (int k = 0; k < n; ++k) {
            h[2 * k] = cos(2 * M_PI * k / n);
            h[2 * k + 1] = -sin(2 * M_PI * k / n);
        }
        CK(hipMemcpy(tw, h.data(), n * 16, hipMemcpyHostToDevice));
        std::vector<double> x(1 << 20);
        for (auto& v : x) v = ((double)rand() / RAND_MAX * 2 - 1) * 1e-3;
        for (long long off = 0; off < N * 2; off += (1 << 20)) CK(hipMemcpy((double*)a + off, x.data(), (1 << 20) * 8, hipMemcpyHostToDevice));
    }
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto mk = [&](int cols, const void* in, void* out, AxisMap im, AxisMap om, TileMap it, TileMap ot, long long ntiles, int tpa) {
        FftLaunch L;
        memset(&L, 0, sizeof(L));
        L.dtype = F64; L.n = n; L.dir = 1; L.cols = cols; L.in = in; L.out = out; L.tw = tw;
        L.imap = im; L.omap = om; L.itile = it; L.otile = ot; L.ntiles = ntiles; L.tiles_per_a = tpa; L.ncols = n;
        return L;
    };
    FftLaunch LZ = mk(0, a, b, plain_axis(n, 1, 0), plain_axis(n, 1, 0), TileMap{n, 0}, TileMap{n, 0}, nn, 1);
    FftLaunch LY = mk(1, b, b, plain_axis(n, n, 1), plain_axis(n, n, 1), TileMap{nn, 1}, TileMap{nn, 1}, nn / 8, n / 8);
    FftLaunch LX = mk(1, a, b, plain_axis(n, nn, 1), plain_axis(n, 1, n), TileMap{n, 1}, TileMap{nn, n}, nn / 8, n / 8);
    struct Case { std::string name; std::function<hipError_t()> run; };
    std::vector<Case> cases;
    //                                                           OSTAGE NT   W  PREFETCH
    cases.push_back({"Z rows           ", [&] { return launch_variant<double2, P512, 1, 4, 1, false, Tune<false, false, 0, false>>(LZ, s); }});
    cases.push_back({"Z rows  prefetch ", [&] { return launch_variant<double2, P512, 1, 4, 1, false, Tune<false, false, 0, true>>(LZ, s); }});
    cases.push_back({"Z rows  prefetch G2", [&] { return launch_variant<double2, P512, 1, 2, 1, false, Tune<false, false, 0, true>>(LZ, s); }});
    cases.push_back({"Y cols           ", [&] { return launch_variant<double2, P512, 8, 1, 1, false, Tune<false, false, 0, false>>(LY, s); }});
    cases.push_back({"Y cols  prefetch ", [&] { return launch_variant<double2, P512, 8, 1, 1, false, Tune<false, false, 0, true>>(LY, s); }});
    cases.push_back({"Y cols  prefetch w4", [&] { return launch_variant<double2, P512, 8, 1, 1, false, Tune<false, false, 4, true>>(LY, s); }});
    cases.push_back({"X cols staged w4 nt          ", [&] { return launch_variant<double2, P512, 8, 1, 1, false, Tune<true, true, 4, false>>(LX, s); }});
    cases.push_back({"X cols staged w4 nt prefetch ", [&] { return launch_variant<double2, P512, 8, 1, 1, false, Tune<true, true, 4, true>>(LX, s); }});
    cases.push_back({"X cols staged w0 nt prefetch ", [&] { return launch_variant<double2, P512, 8, 1, 1, false, Tune<true, true, 0, true>>(LX, s); }});
    cases.push_back({"X cols direct w0 nt prefetch ", [&] { return launch_variant<double2, P512, 8, 1, 1, false, Tune<false, true, 0, true>>(LX, s); }});
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<std::vector<float>> ms(cases.size());
    for (int r = 0; r < rounds + 1; ++r)
        for (size_t i = 0; i < cases.size(); ++i) {
            CK(hipEventRecord(e0, s));
            hipError_t e = cases[i].run();
            if (e != hipSuccess) { printf("%s failed: %s\n", cases[i].name.c_str(), hipGetErrorString(e)); (void)hipGetLastError(); continue; }
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
        printf("%-34s median %.3f ms  %.0f GB/s\n", cases[i].name.c_str(), v[v.size() / 2], 2.0 * 16 * N / v[v.size() / 2] / 1e6);
    }
    return 0;
}
