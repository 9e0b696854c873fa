// tools/kbench6.hip -- developer experiment: variants of the Y-column kernel when its input is Infinity-Cache resident
// (the library's chunked Z+Y stage: Z rows a->b per 64-plane chunk, then Y columns in place on the chunk), 512^3 fp64.
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

template <bool NTL_, bool NTS_, int MINW_, bool PF_> struct Tune {
    static constexpr bool TWPOW = true;
    static constexpr bool OSTAGE = false;
    static constexpr bool NTL = NTL_;
    static constexpr bool NTS = NTS_;
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
    auto mk = [&](int cols, const void* in, void* out, AxisMap im, AxisMap om, TileMap it, TileMap ot, int tpa) {
        FftLaunch L;
        memset(&L, 0, sizeof(L));
        L.dtype = F64; L.n = n; L.dir = 1; L.cols = cols; L.in = in; L.out = out; L.tw = tw;
        L.imap = im; L.omap = om; L.itile = it; L.otile = ot; L.tiles_per_a = tpa; L.ncols = n;
        return L;
    };
    FftLaunch LZ = mk(0, a, b, plain_axis(n, 1, 0), plain_axis(n, 1, 0), TileMap{n, 0}, TileMap{n, 0}, 1);
    FftLaunch LY8 = mk(1, b, b, plain_axis(n, n, 1), plain_axis(n, n, 1), TileMap{nn, 1}, TileMap{nn, 1}, n / 8);
    FftLaunch LY16 = LY8;
    LY16.tiles_per_a = n / 16;
    using TZ = Tune<true, false, 0, false>;
    struct Case { std::string name; std::function<hipError_t(const FftLaunch&)> ylaunch; int cb; int planes; int zg; };
    std::vector<Case> cases;
    auto Y = [&](auto tune, auto cbc) {
        return [&](const FftLaunch& y) { return launch_variant<double2, P512, decltype(cbc)::value, 1, 1, false, decltype(tune)>(y, s); };
    };
    using I8 = std::integral_constant<int, 8>;
    using I16 = std::integral_constant<int, 16>;
    cases.push_back({"Y cb8 prefetch (library)       64 planes", Y(Tune<false, false, 0, true>{}, I8{}), 8, 64, 4});
    cases.push_back({"Y cb8 no prefetch              64 planes", Y(Tune<false, false, 0, false>{}, I8{}), 8, 64, 4});
    cases.push_back({"Y cb8 no prefetch w4 (2 blk)   64 planes", Y(Tune<false, false, 4, false>{}, I8{}), 8, 64, 4});
    cases.push_back({"Y cb16 no prefetch             64 planes", Y(Tune<false, false, 0, false>{}, I16{}), 16, 64, 4});
    cases.push_back({"Y cb8 prefetch nts             64 planes", Y(Tune<false, true, 0, true>{}, I8{}), 8, 64, 4});
    cases.push_back({"Y cb8 prefetch                 48 planes", Y(Tune<false, false, 0, true>{}, I8{}), 8, 48, 4});
    cases.push_back({"Y cb8 prefetch                 56 planes", Y(Tune<false, false, 0, true>{}, I8{}), 8, 56, 4});
    cases.push_back({"Y cb8 prefetch                 32 planes", Y(Tune<false, false, 0, true>{}, I8{}), 8, 32, 4});
    cases.push_back({"Y cb8 prefetch, Z with G2      64 planes", Y(Tune<false, false, 0, true>{}, I8{}), 8, 64, 2});

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<std::vector<float>> ms(cases.size());
    for (int r = 0; r < rounds + 1; ++r)
        for (size_t i = 0; i < cases.size(); ++i) {
            const Case& c = cases[i];
            CK(hipEventRecord(e0, s));
            for (int x0 = 0; x0 < n; x0 += c.planes) {
                const int np = std::min(c.planes, n - x0);
                FftLaunch z = LZ, y = c.cb == 8 ? LY8 : LY16;
                z.a_first = (long long)x0 * n; z.ntiles = (long long)np * n;
                y.a_first = x0; y.ntiles = (long long)np * (n / c.cb);
                if (c.zg == 4) CK((launch_variant<double2, P512, 1, 4, 1, false, TZ>(z, s)));
                else CK((launch_variant<double2, P512, 1, 2, 1, false, TZ>(z, s)));
                CK(c.ylaunch(y));
            }
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float t;
            CK(hipEventElapsedTime(&t, e0, e1));
            if (r > 0) ms[i].push_back(t);
        }
    for (size_t i = 0; i < cases.size(); ++i) {
        auto v = ms[i];
        std::sort(v.begin(), v.end());
        printf("%-48s median %.3f ms\n", cases[i].name.c_str(), v[v.size() / 2]);
    }
    return 0;
}
