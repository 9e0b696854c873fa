// tools/kbench7.hip -- developer experiment: how fast are the row / column kernels when their working set is tiny
// (L2 / Infinity-Cache resident)?  Separates "kernel-structure bound" from "memory bound" for the cache-blocked Z+Y stage.
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

using P512 = Plan<512, 8, 8, 8, 8>;
static AxisMap plain_axis(long long n, long long stride, long long cstride) { return AxisMap{(int)n, 1, 0, stride, cstride, 0}; }

int main(int argc, char** argv) {
    const int n = 512, rounds = argc > 1 ? atoi(argv[1]) : 5;
    const long long nn = (long long)n * n;
    const long long maxplanes = 64;
    double2 *a, *tw;
    CK(hipMalloc(&a, maxplanes * nn * 16));
    CK(hipMalloc(&tw, n * 16));
    {
        std::vector<double> h(2 * (size_t)n);
        for (int k = 0; k < n; ++k) {
            h[2 * k] = cos(2 * M_PI * k / n);
            h[2 * k + 1] = -sin(2 * M_PI * k / n);
        }
        CK(hipMemcpy(tw, h.data(), n * 16, hipMemcpyHostToDevice));
        CK(hipMemset(a, 0, maxplanes * nn * 16));
    }
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int planes : {2, 4, 8, 16, 32, 64}) {
        FftLaunch z;
        memset(&z, 0, sizeof(z));
        z.dtype = F64; z.n = n; z.dir = 1; z.cols = 0; z.in = a; z.out = a; z.tw = tw;
        z.imap = z.omap = plain_axis(n, 1, 0);
        z.itile = z.otile = TileMap{n, 0};
        z.tiles_per_a = 1; z.ncols = 1; z.ntiles = (long long)planes * n;
        FftLaunch y;
        memset(&y, 0, sizeof(y));
        y.dtype = F64; y.n = n; y.dir = 1; y.cols = 1; y.in = a; y.out = a; y.tw = tw;
        y.imap = y.omap = plain_axis(n, n, 1);
        y.itile = y.otile = TileMap{nn, 1};
        y.tiles_per_a = n / 8; y.ncols = n; y.ntiles = (long long)planes * (n / 8);
        const int reps = 2048 / planes;  // always 2048 planes worth of work
        for (int which = 0; which < 2; ++which) {
            std::vector<float> t;
            for (int r = 0; r < rounds + 1; ++r) {
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < reps; ++i) {
                    if (which == 0) CK((launch_variant<double2, P512, 1, 4, 1, false, TuneDefault>(z, s)));
                    else CK((launch_variant<double2, P512, 8, 1, 1, false, TuneCols>(y, s)));
                }
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (r > 0) t.push_back(ms);
            }
            std::sort(t.begin(), t.end());
            const double bytes = 2.0 * 16 * 2048 * nn;
            printf("%s  working set %4d MiB x %4d launches: %.3f ms per 8 GiB of R+W = %.0f GB/s (%.1f us per launch)\n",
                   which == 0 ? "Z rows" : "Y cols", planes * 4, reps, t[t.size() / 2], bytes / t[t.size() / 2] / 1e6,
                   t[t.size() / 2] * 1e3 / reps);
        }
    }
    return 0;
}
