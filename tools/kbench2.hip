// tools/kbench2.hip -- developer experiment (not part of the library): overlap the HBM-bound Z pass of chunk k+1 with the
// cache-bound Y pass of chunk k on two streams; and wider staged X tiles.  512^3 fp64.
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

template <bool OSTAGE_, bool NTL_, bool NTS_, int MINW_> struct Tune {
    static constexpr bool TWPOW = true;
    static constexpr bool OSTAGE = OSTAGE_;
    static constexpr bool NTL = NTL_;
    static constexpr bool NTS = NTS_;
    static constexpr int MIN_WAVES = MINW_;
    static constexpr int CB_OVERRIDE = 0;
    static constexpr bool PLAIN = false;
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

int main(int argc, char** argv) {
    const int n = 512;
    const int rounds = argc > 1 ? atoi(argv[1]) : 7;
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
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    std::vector<hipEvent_t> evz(64), evy(64);
    for (auto& e : evz) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : evy) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));

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
        return L;
    };
    FftLaunch LZ = mk(0, a, b, plain_axis(n, 1, 0), plain_axis(n, 1, 0), TileMap{n, 0}, TileMap{n, 0}, nn, 1);      // a -> b
    FftLaunch LY = mk(1, b, b, plain_axis(n, n, 1), plain_axis(n, n, 1), TileMap{nn, 1}, TileMap{nn, 1}, nn / 8, n / 8);
    FftLaunch LX = mk(1, a, b, plain_axis(n, nn, 1), plain_axis(n, 1, n), TileMap{n, 1}, TileMap{nn, n}, nn / 8, n / 8);
    FftLaunch LX16 = LX;
    LX16.ntiles = nn / 16;
    LX16.tiles_per_a = n / 16;

    using TZ = Tune<false, true, false, 0>;   // stream-in rows
    using TY = Tune<false, false, false, 0>;  // plain cols
    struct Case {
        std::string name;
        std::function<hipError_t()> run;  // enqueues on sa (and sb), must end with everything joined into sa
    };
    std::vector<Case> cases;

    auto serial = [&](int planes) {
        return [&, planes]() -> hipError_t {
            for (int x0 = 0; x0 < n; x0 += planes) {
                FftLaunch z = LZ, y = LY;
                z.a_first = (long long)x0 * n;
                z.ntiles = (long long)planes * n;
                y.a_first = x0;
                y.ntiles = (long long)planes * (n / 8);
                hipError_t e = launch_variant<double2, P512, 1, 4, 1, false, TZ>(z, sa);
                if (e != hipSuccess) return e;
                e = launch_variant<double2, P512, 8, 1, 1, false, TY>(y, sa);
                if (e != hipSuccess) return e;
            }
            return hipSuccess;
        };
    };
    // Z on stream A (capped at zb blocks/CU), Y on stream B; Z(k) may not start before Y(k - lag) finished (cache budget)
    auto overlapped = [&](int planes, int zb, int lag) {
        return [&, planes, zb, lag]() -> hipError_t {
            const int nch = n / planes;
            for (int k = 0; k < nch; ++k) {
                FftLaunch z = LZ, y = LY;
                z.a_first = (long long)k * planes * n;
                z.ntiles = (long long)planes * n;
                z.blocks_per_cu_limit = zb;
                y.a_first = (long long)k * planes;
                y.ntiles = (long long)planes * (n / 8);
                if (k - lag >= 0) CK(hipStreamWaitEvent(sa, evy[k - lag], 0));
                hipError_t e = launch_variant<double2, P512, 1, 4, 1, false, TZ>(z, sa);
                if (e != hipSuccess) return e;
                CK(hipEventRecord(evz[k], sa));
                CK(hipStreamWaitEvent(sb, evz[k], 0));
                e = launch_variant<double2, P512, 8, 1, 1, false, TY>(y, sb);
                if (e != hipSuccess) return e;
                CK(hipEventRecord(evy[k], sb));
            }
            CK(hipStreamWaitEvent(sa, evy[nch - 1], 0));
            return hipSuccess;
        };
    };
    cases.push_back({"Z+Y serial, 64-plane chunks (library default)", serial(64)});
    cases.push_back({"Z+Y serial, 32-plane chunks", serial(32)});
    for (int planes : {16, 32, 64})
        for (int zb : {1, 2})
            for (int lag : {1, 2}) {
                if (planes * lag > 64 + 32) continue;
                cases.push_back({"Z||Y two streams, " + std::to_string(planes) + " planes, Z<=" + std::to_string(zb) +
                                     " blk/CU, lag " + std::to_string(lag),
                                 overlapped(planes, zb, lag)});
            }
    cases.push_back({"X cols cb8 staged w4 nt (library)", [&]() { return launch_variant<double2, P512, 8, 1, 1, false, Tune<true, true, true, 4>>(LX, sa); }});
    cases.push_back({"X cols cb16 staged w4 nt", [&]() { return launch_variant<double2, P512, 16, 1, 1, false, Tune<true, true, true, 4>>(LX16, sa); }});
    cases.push_back({"X cols cb16 staged w0 nt", [&]() { return launch_variant<double2, P512, 16, 1, 1, false, Tune<true, true, true, 0>>(LX16, sa); }});

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<std::vector<float>> ms(cases.size());
    for (int r = 0; r < rounds + 1; ++r)
        for (size_t i = 0; i < cases.size(); ++i) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, sa));
            hipError_t e = cases[i].run();
            if (e != hipSuccess) {
                printf("%s: failed %s\n", cases[i].name.c_str(), hipGetErrorString(e));
                (void)hipGetLastError();
                continue;
            }
            CK(hipEventRecord(e1, sa));
            CK(hipEventSynchronize(e1));
            float t;
            CK(hipEventElapsedTime(&t, e0, e1));
            if (r > 0) ms[i].push_back(t);
        }
    for (size_t i = 0; i < cases.size(); ++i) {
        auto v = ms[i];
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        printf("%-58s median %.3f ms  min %.3f ms\n", cases[i].name.c_str(), v[v.size() / 2], v[0]);
    }
    return 0;
}
