// tools/kbench7.hip -- developer experiment: does a small REUSED scratch between the Z and Y passes keep the intermediate
// out of HBM altogether?  (library: Z a->b chunk, Y on the chunk of b: the dirty chunk of b is eventually written back by
// the Infinity Cache.  Here: Z a->scratch, Y scratch->b; the scratch lines are overwritten while still cached.)  512^3 fp64.
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
    double2* sc;
    CK(hipMalloc(&sc, 2 * 64 * nn * 16));
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
    FftLaunch LY = mk(1, b, b, plain_axis(n, n, 1), plain_axis(n, n, 1), TileMap{nn, 1}, TileMap{nn, 1}, n / 8);
    using TZ = Tune<true, false, 0, false>;    // Z: streaming loads, cached stores
    using TZS = Tune<true, true, 0, false>;    // Z: streaming loads and stores
    using TY = Tune<false, false, 0, true>;    // Y as in the library (in place)
    using TYS = Tune<false, true, 0, true>;    // Y with streaming stores
    using TYLS = Tune<true, true, 0, true>;    // Y with streaming loads and stores
    // mode 0: library (Z a->b, Y b->b).  1: Z a->scratch, Y scratch->b (one scratch).  2: two alternating scratches.
    // 3: like 1 with nt loads in Y.  4: like 1 with nt stores in Z too.  5: Z a->b, Y b->c out of place streaming (P>1 shape)
    struct Case { const char* name; int mode; int planes; };
    std::vector<Case> cases = {
        {"library: Z a->b, Y in place           64 pl", 0, 64}, {"library: Z a->b, Y in place           32 pl", 0, 32},
        {"Z a->b, Y b->a(out of place, nts)     64 pl", 5, 64},
        {"scratch x1, Y nts                     64 pl", 1, 64}, {"scratch x1, Y nts                     48 pl", 1, 48},
        {"scratch x1, Y nts                     32 pl", 1, 32}, {"scratch x1, Y nts                     24 pl", 1, 24},
        {"scratch x1, Y nts                     16 pl", 1, 16}, {"scratch x1, Y nts                      8 pl", 1, 8},
        {"scratch x2, Y nts                     32 pl", 2, 32}, {"scratch x2, Y nts                     16 pl", 2, 16},
        {"scratch x2, Y nts                      8 pl", 2, 8},
        {"scratch x1, Y ntl+nts                 32 pl", 3, 32}, {"scratch x1, Y ntl+nts                 16 pl", 3, 16},
        {"scratch x1, Z nts, Y nts              32 pl", 4, 32},
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<std::vector<float>> ms(cases.size());
    for (int r = 0; r < rounds + 1; ++r)
        for (size_t i = 0; i < cases.size(); ++i) {
            const Case& c = cases[i];
            CK(hipEventRecord(e0, s));
            int chunk = 0;
            for (int x0 = 0; x0 < n; x0 += c.planes, ++chunk) {
                const int np = std::min(c.planes, n - x0);
                FftLaunch z = LZ, y = LY;
                z.a_first = (long long)x0 * n; z.ntiles = (long long)np * n;
                y.a_first = x0; y.ntiles = (long long)np * (n / 8);
                if (c.mode >= 1 && c.mode <= 4) {
                    double2* sbase = sc + (c.mode == 2 ? (long long)(chunk & 1) * c.planes * nn : 0) - (long long)x0 * nn;
                    z.out = sbase;
                    y.in = sbase;
                }
                if (c.mode == 5) y.out = a;
                if (c.mode == 4) CK((launch_variant<double2, P512, 1, 4, 1, false, TZS>(z, s)));
                else CK((launch_variant<double2, P512, 1, 4, 1, false, TZ>(z, s)));
                if (c.mode == 0) CK((launch_variant<double2, P512, 8, 1, 1, false, TY>(y, s)));
                else if (c.mode == 3) CK((launch_variant<double2, P512, 8, 1, 1, false, TYLS>(y, s)));
                else CK((launch_variant<double2, P512, 8, 1, 1, false, TYS>(y, s)));
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
        printf("%-48s median %.3f ms\n", cases[i].name, v[v.size() / 2]);
    }
    return 0;
}
