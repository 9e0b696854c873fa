// tools/kbench.hip -- developer A/B bench of kernel tuning variants (not part of the library, not graded).
// Instantiates the FFT kernel template with different Tune policies and times the passes of the 512^3 fp64
// single-GPU pipeline with HIP events, interleaved rounds, median reported.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I distributedfft_amd/csrc -I include tools/kbench.hip -o tools/bin/kbench
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

template <bool OSTAGE_, bool NTL_, bool NTS_, int MINW_, bool PLAIN_> struct Tune {
    static constexpr bool TWPOW = true;
    static constexpr bool OSTAGE = OSTAGE_;
    static constexpr bool NTL = NTL_;
    static constexpr bool NTS = NTS_;
    static constexpr int MIN_WAVES = MINW_;
    static constexpr int CB_OVERRIDE = 0;
    static constexpr bool PLAIN = PLAIN_;
    static constexpr bool PREFETCH = false;
};

using P512 = Plan<512, 8, 8, 8, 8>;

static AxisMap plain_axis(long long n, long long stride, long long cstride) {
    AxisMap m;
    m.blk = (int)n;
    m.nblk = 1;
    m.blk_stride = 0;
    m.stride = stride;
    m.cstride = cstride;
    m.last_delta = 0;
    return m;
}

struct Case {
    std::string name;
    std::function<hipError_t(hipStream_t)> run;
    double passes = 1;
};

int main(int argc, char** argv) {
    const int n = 512;
    const int rounds = argc > 1 ? atoi(argv[1]) : 7;
    const long long N = (long long)n * n * n;
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
    CK(hipStreamCreate(&s));

    auto mk = [&](int cols, const void* in, void* out, AxisMap im, AxisMap om, TileMap it, TileMap ot, long long ntiles, int tpa) {
        FftLaunch L;
        memset(&L, 0, sizeof(L));
        L.dtype = F64;
        L.n = n;
        L.dir = 1;
        L.cols = cols;
        L.in = in;
        L.out = out;
        L.tw = tw;
        L.imap = im;
        L.omap = om;
        L.itile = it;
        L.otile = ot;
        L.ntiles = ntiles;
        L.tiles_per_a = tpa;
        L.ncols = n;
        L.a_first = 0;
        return L;
    };
    const long long nn = (long long)n * n;
    FftLaunch LZ = mk(0, a, a, plain_axis(n, 1, 0), plain_axis(n, 1, 0), TileMap{n, 0}, TileMap{n, 0}, nn, 1);
    FftLaunch LZo = LZ;  // out of place a -> b (the bench's INPUT_FROM_IN mode)
    LZo.out = b;
    FftLaunch LY = mk(1, a, a, plain_axis(n, n, 1), plain_axis(n, n, 1), TileMap{nn, 1}, TileMap{nn, 1}, nn / 8, n / 8);
    FftLaunch LX = mk(1, a, b, plain_axis(n, nn, 1), plain_axis(n, 1, n), TileMap{n, 1}, TileMap{nn, n}, nn / 8, n / 8);

    std::vector<Case> cases;
#define ADD(NAME, PASSES, EXPR)                                                \
    {                                                                          \
        Case c_;                                                               \
        c_.name = NAME;                                                        \
        c_.passes = PASSES;                                                    \
        c_.run = [&](hipStream_t st) -> hipError_t { return EXPR; };           \
        cases.push_back(c_);                                                   \
    }
    //            OSTAGE NTL    NTS    W  PLAIN
    using T0 = Tune<false, false, false, 0, false>;
    using TP = Tune<false, false, false, 0, true>;
    using TPW = Tune<false, false, false, 4, true>;
    using TPW3 = Tune<false, false, false, 3, true>;
    using TL = Tune<false, true, false, 0, false>;
    using TS = Tune<false, false, true, 0, false>;
    using TLS = Tune<false, true, true, 0, false>;
    using TPLS = Tune<false, true, true, 0, true>;
    using TPWLS = Tune<false, true, true, 4, true>;
    using TO4 = Tune<true, false, false, 4, false>;
    using TO4L = Tune<true, true, false, 4, false>;
    using TO4S = Tune<true, false, true, 4, false>;
    using TO4LS = Tune<true, true, true, 4, false>;
    using TO3LS = Tune<true, true, true, 3, false>;
    using TO0LS = Tune<true, true, true, 0, false>;

    ADD("Z rows G4 base", 1, (launch_variant<double2, P512, 1, 4, 1, false, T0>(LZ, st)));
    ADD("Z rows G4 plain-addr", 1, (launch_variant<double2, P512, 1, 4, 1, false, TP>(LZ, st)));
    ADD("Z rows G4 plain-addr w4", 1, (launch_variant<double2, P512, 1, 4, 1, false, TPW>(LZ, st)));
    ADD("Z rows G4 plain-addr w3", 1, (launch_variant<double2, P512, 1, 4, 1, false, TPW3>(LZ, st)));
    ADD("Z rows G4 ntl", 1, (launch_variant<double2, P512, 1, 4, 1, false, TL>(LZ, st)));
    ADD("Z rows G4 nts", 1, (launch_variant<double2, P512, 1, 4, 1, false, TS>(LZ, st)));
    ADD("Z rows G4 ntl+nts", 1, (launch_variant<double2, P512, 1, 4, 1, false, TLS>(LZ, st)));
    ADD("Z rows G4 plain-addr ntl+nts", 1, (launch_variant<double2, P512, 1, 4, 1, false, TPLS>(LZ, st)));
    ADD("Z rows G4 plain-addr w4 ntl+nts", 1, (launch_variant<double2, P512, 1, 4, 1, false, TPWLS>(LZ, st)));
    ADD("Z rows G2 plain-addr ntl+nts", 1, (launch_variant<double2, P512, 1, 2, 1, false, TPLS>(LZ, st)));
    ADD("Z rows G4 ntl+nts out-of-place a->b", 1, (launch_variant<double2, P512, 1, 4, 1, false, TLS>(LZo, st)));
    ADD("Y cols base", 1, (launch_variant<double2, P512, 8, 1, 1, false, T0>(LY, st)));
    ADD("Y cols plain-addr", 1, (launch_variant<double2, P512, 8, 1, 1, false, TP>(LY, st)));
    ADD("Y cols plain-addr w4", 1, (launch_variant<double2, P512, 8, 1, 1, false, TPW>(LY, st)));
    ADD("Y cols ntl+nts", 1, (launch_variant<double2, P512, 8, 1, 1, false, TLS>(LY, st)));
    ADD("Y cols plain-addr ntl+nts", 1, (launch_variant<double2, P512, 8, 1, 1, false, TPLS>(LY, st)));
    ADD("Y cols plain-addr w4 ntl+nts", 1, (launch_variant<double2, P512, 8, 1, 1, false, TPWLS>(LY, st)));
    ADD("X cols base", 1, (launch_variant<double2, P512, 8, 1, 1, false, T0>(LX, st)));
    ADD("X cols plain-addr w4 ntl+nts", 1, (launch_variant<double2, P512, 8, 1, 1, false, TPWLS>(LX, st)));
    ADD("X cols ostage w4", 1, (launch_variant<double2, P512, 8, 1, 1, false, TO4>(LX, st)));
    ADD("X cols ostage w4 ntl", 1, (launch_variant<double2, P512, 8, 1, 1, false, TO4L>(LX, st)));
    ADD("X cols ostage w4 nts", 1, (launch_variant<double2, P512, 8, 1, 1, false, TO4S>(LX, st)));
    ADD("X cols ostage w4 ntl+nts", 1, (launch_variant<double2, P512, 8, 1, 1, false, TO4LS>(LX, st)));
    ADD("X cols ostage w3 ntl+nts", 1, (launch_variant<double2, P512, 8, 1, 1, false, TO3LS>(LX, st)));
    ADD("X cols ostage w0 ntl+nts", 1, (launch_variant<double2, P512, 8, 1, 1, false, TO0LS>(LX, st)));

    // chunked Z+Y: both passes over a chunk of planes before moving on, so the Y pass hits the 256 MiB Infinity Cache.
    // Z: non-temporal loads (input is streamed once), plain stores (keep the chunk cached);
    // Y: plain loads (cache hits), non-temporal stores (streamed out).
    for (int planes : {32, 48, 64, 96, 128}) {
        Case c_;
        c_.name = "Z+Y chunked " + std::to_string(planes) + " planes (" + std::to_string(planes * 4) + " MiB) Z:ntl Y:nts";
        c_.passes = 2;
        c_.run = [&, planes](hipStream_t st) -> hipError_t {
            for (int x0 = 0; x0 < n; x0 += planes) {
                const int np = std::min(planes, n - x0);
                FftLaunch z = LZ, y = LY;
                z.a_first = (long long)x0 * n;
                z.ntiles = (long long)np * n;
                y.a_first = x0;
                y.ntiles = (long long)np * (n / 8);
                hipError_t e = launch_variant<double2, P512, 1, 4, 1, false, TL>(z, st);
                if (e != hipSuccess) return e;
                e = launch_variant<double2, P512, 8, 1, 1, false, TS>(y, st);
                if (e != hipSuccess) return e;
            }
            return hipSuccess;
        };
        cases.push_back(c_);
    }
    for (int planes : {64}) {
        Case c_;
        c_.name = "Z+Y chunked " + std::to_string(planes) + " planes, all plain";
        c_.passes = 2;
        c_.run = [&, planes](hipStream_t st) -> hipError_t {
            for (int x0 = 0; x0 < n; x0 += planes) {
                FftLaunch z = LZ, y = LY;
                z.a_first = (long long)x0 * n;
                z.ntiles = (long long)planes * n;
                y.a_first = x0;
                y.ntiles = (long long)planes * (n / 8);
                hipError_t e = launch_variant<double2, P512, 1, 4, 1, false, T0>(z, st);
                if (e != hipSuccess) return e;
                e = launch_variant<double2, P512, 8, 1, 1, false, T0>(y, st);
                if (e != hipSuccess) return e;
            }
            return hipSuccess;
        };
        cases.push_back(c_);
        Case d_;
        d_.name = "Z(a->b)+Y(b) chunked " + std::to_string(planes) + " planes Z:ntl Y:nts (bench mode)";
        d_.passes = 2;
        d_.run = [&, planes](hipStream_t st) -> hipError_t {
            for (int x0 = 0; x0 < n; x0 += planes) {
                FftLaunch z = LZo, y = LY;
                z.a_first = (long long)x0 * n;
                z.ntiles = (long long)planes * n;
                y.in = b;
                y.out = b;
                y.a_first = x0;
                y.ntiles = (long long)planes * (n / 8);
                hipError_t e = launch_variant<double2, P512, 1, 4, 1, false, TL>(z, st);
                if (e != hipSuccess) return e;
                e = launch_variant<double2, P512, 8, 1, 1, false, TS>(y, st);
                if (e != hipSuccess) return e;
            }
            return hipSuccess;
        };
        cases.push_back(d_);
    }
    ADD("Z+Y whole slab, ntl+nts both", 2, ([&]() -> hipError_t {
            hipError_t e = launch_variant<double2, P512, 1, 4, 1, false, TLS>(LZ, st);
            if (e != hipSuccess) return e;
            return launch_variant<double2, P512, 8, 1, 1, false, TLS>(LY, st);
        })());
    ADD("hipMemcpyAsync D2D (a->b)", 1, hipMemcpyAsync(b, a, N * 16, hipMemcpyDeviceToDevice, st));

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<std::vector<float>> ms(cases.size());
    for (int r = 0; r < rounds + 1; ++r) {
        for (size_t i = 0; i < cases.size(); ++i) {
            CK(hipEventRecord(e0, s));
            hipError_t e = cases[i].run(s);
            if (e != hipSuccess) {
                printf("%s: launch failed: %s\n", cases[i].name.c_str(), hipGetErrorString(e));
                (void)hipGetLastError();
                ms[i].push_back(-1);
                continue;
            }
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float t;
            CK(hipEventElapsedTime(&t, e0, e1));
            if (r > 0) ms[i].push_back(t);
        }
    }
    const double bytes = 2.0 * 16 * N;
    for (size_t i = 0; i < cases.size(); ++i) {
        auto v = ms[i];
        if (v.empty() || v[0] < 0) continue;
        std::sort(v.begin(), v.end());
        printf("%-62s median %.3f ms  min %.3f ms  %.0f GB/s algorithmic\n", cases[i].name.c_str(), v[v.size() / 2], v[0],
               cases[i].passes * bytes / v[v.size() / 2] / 1e6);
    }
    return 0;
}
