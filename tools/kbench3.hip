// tools/kbench3.hip -- developer measurement (not part of the library): the LOCAL work of one rank of the 512^3 fp64
// problem at P = 1, 2, 4, 8 (slab [512/P][512][512] in, [512][512/P][512] after the exchange), timed on one GPU without any
// exchange: Z+Y (packed store) and X (staged transposed store) exactly as the library launches them.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
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

static AxisMap plain_axis(long long n, long long stride, long long cstride) {
    AxisMap m{(int)n, 1, 0, stride, cstride, 0};
    return m;
}

int main(int argc, char** argv) {
    const int n = 512, rounds = argc > 1 ? atoi(argv[1]) : 9;
    const long long N = (long long)n * n * n, nn = (long long)n * n;
    double2 *a, *b, *c, *tw;
    CK(hipMalloc(&a, N * 16));
    CK(hipMalloc(&b, N * 16));
    CK(hipMalloc(&c, N * 16));
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
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventCreate(&e2));

    for (int P : {1, 2, 4, 8}) {
        const long long xs = n / P, ys = n / P;
        std::vector<float> tzy, tx;
        for (int r = 0; r < rounds + 1; ++r) {
            FftLaunch z;
            memset(&z, 0, sizeof(z));
            z.dtype = F64; z.n = n; z.dir = 1; z.cols = 0; z.in = a; z.out = b; z.tw = tw;
            z.imap = z.omap = plain_axis(n, 1, 0);
            z.itile = z.otile = TileMap{n, 0};
            z.tiles_per_a = 1; z.ncols = 1;
            FftLaunch y;
            memset(&y, 0, sizeof(y));
            y.dtype = F64; y.n = n; y.dir = 1; y.cols = 1; y.in = b; y.out = (P > 1) ? (void*)c : (void*)b; y.tw = tw;
            y.imap = plain_axis(n, n, 1);
            y.itile = TileMap{nn, 1};
            if (P > 1) {
                y.omap = AxisMap{(int)ys, P, xs * ys * n, (long long)n, 1, 0};
                y.otile = TileMap{ys * n, 1};
            } else {
                y.omap = y.imap;
                y.otile = y.itile;
            }
            y.tiles_per_a = n / 8; y.ncols = n;
            const long long cp = std::min<long long>(64, xs);
            const bool chunked = cp < xs;
            CK(hipEventRecord(e0, s));
            for (long long x0 = 0; x0 < xs; x0 += cp) {
                z.a_first = x0 * n; z.ntiles = cp * n; z.hints = chunked ? FFT_HINT_STREAM_IN : 0;
                y.a_first = x0; y.ntiles = cp * (n / 8); y.hints = (chunked && P > 1) ? FFT_HINT_STREAM_OUT : 0;
                CK((launch_plan<double2, P512>(z, s)));
                CK((launch_plan<double2, P512>(y, s)));
            }
            CK(hipEventRecord(e1, s));
            // X pass on a [512][ys][512] slab (content irrelevant for timing): a -> b
            FftLaunch x;
            memset(&x, 0, sizeof(x));
            x.dtype = F64; x.n = n; x.dir = 1; x.cols = 1; x.in = a; x.out = b; x.tw = tw;
            x.imap = plain_axis(n, ys * n, 1);
            x.itile = TileMap{n, 1};
            x.omap = plain_axis(n, 1, n);
            x.otile = TileMap{nn, n};
            x.tiles_per_a = n / 8; x.ntiles = ys * (n / 8); x.ncols = n;
            CK((launch_plan<double2, P512>(x, s)));
            CK(hipEventRecord(e2, s));
            CK(hipEventSynchronize(e2));
            float t1, t2;
            CK(hipEventElapsedTime(&t1, e0, e1));
            CK(hipEventElapsedTime(&t2, e1, e2));
            if (r > 0) { tzy.push_back(t1); tx.push_back(t2); }
        }
        std::sort(tzy.begin(), tzy.end());
        std::sort(tx.begin(), tx.end());
        const double bytes = 2.0 * 16 * N / P;
        printf("P=%d local: Z+Y %.3f ms (%.0f GB/s alg, 2 passes)  X %.3f ms (%.0f GB/s)  local total %.3f ms\n", P,
               tzy[tzy.size() / 2], 2 * bytes / tzy[tzy.size() / 2] / 1e6, tx[tx.size() / 2], bytes / tx[tx.size() / 2] / 1e6,
               tzy[tzy.size() / 2] + tx[tx.size() / 2]);
    }
    return 0;
}
