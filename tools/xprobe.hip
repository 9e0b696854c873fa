// tools/xprobe.hip -- developer probe (not part of the library): why does the 512^3 fp64 X pass run at 0.71 ms on some
// hand-over buffers and at 0.77 ms on others?  Launches the library's own X-pass kernel (same template instantiation) on
// controlled (W, out) buffer pairs:
//   A  matrix of NW hipMalloc'ed hand-over buffers x NO output buffers      -> is it W alone, out alone, or the pair?
//   B  plane padding sweep on the slowest and the fastest W of part A       -> does some padding make every buffer fast?
//   C  W assembled from 256 MiB physical chunks (HIP VMM) drawn from a pool -> is "slow" a property of physical chunks?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I distributedfft_amd/csrc -I include tools/xprobe.hip -o tools/bin/xprobe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
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
static const int N = 512;

static AxisMap plain_axis(long long n, long long stride, long long cstride) {
    AxisMap m;
    std::memset(&m, 0, sizeof(m));
    m.blk = (int)n;
    m.nblk = 1;
    m.stride = stride;
    m.cstride = cstride;
    m.sub = 1;
    return m;
}

// plain streaming copy, 16 B per lane, grid-stride (256 x 1024 threads: the best shape of round 2's copy sweep)
typedef double d2v __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(1024) copy_kernel(const d2v* __restrict__ a, d2v* __restrict__ b, size_t n) {
    size_t       i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i + 3 * st < n; i += 4 * st) {
        const d2v v0 = __builtin_nontemporal_load(a + i), v1 = __builtin_nontemporal_load(a + i + st), v2 = __builtin_nontemporal_load(a + i + 2 * st),
                  v3 = __builtin_nontemporal_load(a + i + 3 * st);
        __builtin_nontemporal_store(v0, b + i);
        __builtin_nontemporal_store(v1, b + i + st);
        __builtin_nontemporal_store(v2, b + i + 2 * st);
        __builtin_nontemporal_store(v3, b + i + 3 * st);
    }
    for (; i < n; i += st) b[i] = a[i];
}
static const double2* g_tw = nullptr;
static hipStream_t    g_s;
static hipStream_t    g_s_fwd() { return g_s; }

// layout of W: 0 = [x][y][z] with planes `plane` elements apart (the library's hand-over buffer);
//              1 = [y][zt][x][8]: every X tile is one contiguous 64 KiB run;
//              2 = [xhi][y][zt][xlo][8] with xlo = 64 planes (a cache chunk): every X tile is 8 runs of 8 KiB
static int g_layout = 0;
// X pass: W -> out = [y][z][kx]
static hipError_t xpass(const void* W, void* out, long long plane) {
    FftLaunch L;
    std::memset(&L, 0, sizeof(L));
    L.dtype = F64;
    L.n = N;
    L.dir = 1;
    L.cols = 1;
    L.in = W;
    L.out = out;
    L.tw = g_tw;
    L.imap = plain_axis(N, plane, 1);
    L.itile = TileMap{(long long)N, 1};
    if (g_layout == 1) {
        L.imap = plain_axis(N, 8, 1);
        L.itile = TileMap{(long long)(N / 8) * N * 8, (long long)N};
    } else if (g_layout == 2) {
        L.imap.blk = 64;
        L.imap.nblk = N / 64;
        L.imap.blk_stride = 64ll * N * N;
        L.imap.stride = 8;
        L.itile = TileMap{(long long)(N / 8) * 64 * 8, 64};
    }
    L.omap = plain_axis(N, 1, N);
    L.otile = TileMap{(long long)N * N, (long long)N};
    L.na = N;
    L.ncols = N;
    L.tiles_per_a = N / 8;
    L.ntiles = (long long)N * (N / 8);
    L.scale = 1.0;
    return launch_variant<double2, P512, 8, 1, +1, false, TuneTransposedStore>(L, g_s);
}

static float time_copy(const void* a, void* b, int mode, int warm = 2, int reps = 5) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> t;
    const size_t       n = (size_t)N * N * N;
    for (int i = 0; i < warm + reps; ++i) {
        CK(hipEventRecord(e0, g_s_fwd()));
        if (mode == 0) hipLaunchKernelGGL(copy_kernel, dim3(256), dim3(1024), 0, g_s_fwd(), (const d2v*)a, (d2v*)b, n);
        else CK(hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, g_s_fwd()));
        CK(hipEventRecord(e1, g_s_fwd()));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (i >= warm) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return t[t.size() / 2];
}
static float time_x(const void* W, void* out, long long plane, int warm = 3, int reps = 7) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> t;
    for (int i = 0; i < warm + reps; ++i) {
        CK(hipEventRecord(e0, g_s));
        CK(xpass(W, out, plane));
        CK(hipEventRecord(e1, g_s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (i >= warm) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return t[t.size() / 2];
}

int main(int argc, char** argv) {
    const int NW = argc > 1 ? atoi(argv[1]) : 8, NO = argc > 2 ? atoi(argv[2]) : 3, NPOOL = argc > 3 ? atoi(argv[3]) : 27,  /* unused */
              TRIALS = argc > 4 ? atoi(argv[4]) : 40, SPAN_GB = argc > 5 ? atoi(argv[5]) : 8, STEP_MB = argc > 6 ? atoi(argv[6]) : 256;
    CK(hipStreamCreateWithFlags(&g_s, hipStreamNonBlocking));
    {
        std::vector<double> h(2 * N);
        for (int k = 0; k < N; ++k) {
            h[2 * k] = cos(2 * M_PI * k / N);
            h[2 * k + 1] = -sin(2 * M_PI * k / N);
        }
        void* d;
        CK(hipMalloc(&d, h.size() * 8));
        CK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        g_tw = (const double2*)d;
    }
    if (NW < 0) {
        // census: NO buffers of 2 GiB (+ 8 MiB) allocated one after the other; each is classified by the X pass against the
        // first one, as the buffer read (-> B0) and as the buffer written (<- B0), and against its predecessor
        const size_t    sl = (size_t)N * N * N * 16 + (8u << 20);
        const long long pl = (long long)N * N + 3 * 8;
        std::vector<void*> B(NO);
        for (auto& b : B) {
            CK(hipMalloc(&b, sl));
            CK(hipMemset(b, 0, sl));
        }
        CK(hipDeviceSynchronize());
        printf("== census of %d buffers (2 GiB each): X pass Bi -> B0 | X pass B0 -> Bi | copy kernel Bi -> B0 | copy kernel B0 -> Bi | hipMemcpy Bi -> B0 | copy Bi -> Bi-1\n", NO);
        for (int i = 1; i < NO; ++i) {
            printf("  B%-3d @%p   %.4f | %.4f | %.4f | %.4f | %.4f | %.4f\n", i, B[i], time_x(B[i], B[0], pl, 2, 5), time_x(B[0], B[i], pl, 2, 5),
                   time_copy(B[i], B[0], 0), time_copy(B[0], B[i], 0), time_copy(B[i], B[0], 1), time_copy(B[i], B[i - 1], 0));
            fflush(stdout);
        }
        return 0;
    }
    const size_t    slab = (size_t)N * N * N * 16, extra = 8u << 20;
    const long long plane3 = (long long)N * N + 3 * 8;
    std::vector<void*> W(NW), O(NO);
    for (auto& o : O) {
        CK(hipMalloc(&o, slab));
        CK(hipMemset(o, 0, slab));
    }
    for (auto& w : W) {
        CK(hipMalloc(&w, slab + extra));
        CK(hipMemset(w, 0, slab + extra));
    }
    CK(hipDeviceSynchronize());
    for (int j = 0; j < NO; ++j) printf("   O%d @%p\n", j, O[j]);
    printf("== A: X pass ms, rows = hand-over buffers (hipMalloc, planes + 3 lines), columns = output buffers\n");
    std::vector<float> wmean(NW, 0.f);
    for (int r = 0; r < 2; ++r)  // two rounds: repeatability
        for (int i = 0; i < NW; ++i) {
            printf("  round %d W%d @%p:", r, i, W[i]);
            for (int j = 0; j < NO; ++j) {
                const float t = time_x(W[i], O[j], plane3);
                wmean[i] += t / (2 * NO);
                printf("  %.4f", t);
            }
            printf("\n");
            fflush(stdout);
        }
    int slow = 0, fast = 0;
    for (int i = 0; i < NW; ++i) {
        if (wmean[i] > wmean[slow]) slow = i;
        if (wmean[i] < wmean[fast]) fast = i;
    }
    printf("   slowest W%d %.4f ms, fastest W%d %.4f ms\n", slow, wmean[slow], fast, wmean[fast]);
    printf("== B: plane padding (lines of 128 B) on the slowest / fastest buffer, out0\n");
    for (int pad : {0, 3, 5}) {
        const long long pl = (long long)N * N + pad * 8;
        printf("  pad %2d lines: slow W%d %.4f   fast W%d %.4f\n", pad, slow, time_x(W[slow], O[0], pl), fast, time_x(W[fast], O[0], pl));
        fflush(stdout);
    }
    {
        // fastest and slowest (W, out) pair of part A (last round's numbers are recomputed here)
        int   fw = 0, fo = 0, sw = 0, so = 0;
        float tf = 1e9f, ts = 0.f;
        for (int i = 0; i < NW; ++i)
            for (int j = 0; j < NO; ++j) {
                const float t = time_x(W[i], O[j], plane3, 2, 5);
                if (t < tf) tf = t, fw = i, fo = j;
                if (t > ts) ts = t, sw = i, so = j;
            }
        printf("== E: layout of the hand-over buffer, fast pair (W%d, O%d) | slow pair (W%d, O%d) | both crossed\n", fw, fo, sw, so);
        const char* names[3] = {"[x][y][z] planes + 3 lines (library)", "[y][zt][x][8]: 64 KiB runs", "[xhi][y][zt][xlo=64][8]: 8 KiB runs"};
        for (int r = 0; r < 2; ++r)
            for (int lay = 0; lay < 3; ++lay) {
                g_layout = lay;
                printf("  %-40s  %.4f | %.4f | %.4f  %.4f\n", names[lay], time_x(W[fw], O[fo], plane3), time_x(W[sw], O[so], plane3),
                       time_x(W[fw], O[so], plane3), time_x(W[sw], O[fo], plane3));
            }
        g_layout = 0;
    }
    for (int i = 0; i < NW; ++i) CK(hipFree(W[i]));
    W.clear();
    // ---- D: where inside ONE large allocation does a buffer start? ----
    {
        const size_t span = (size_t)SPAN_GB << 30, step = (size_t)STEP_MB << 20;
        void*        big = nullptr;
        CK(hipMalloc(&big, span + slab + extra));
        CK(hipMemset(big, 0, span + slab + extra));
        void* Wfix[2] = {nullptr, nullptr};
        for (auto& w : Wfix) {
            CK(hipMalloc(&w, slab + extra));
            CK(hipMemset(w, 0, slab + extra));
        }
        CK(hipDeviceSynchronize());
        printf("== D: one allocation of %d GiB + 2 GiB @%p; buffer = 2 GiB window starting k x %d MiB into it\n", SPAN_GB, big, STEP_MB);
        printf("   side W: X pass reads the window, writes O0 / O1;   side out: X pass reads Wfix0 / Wfix1, writes the window\n");
        for (size_t off = 0; off <= span; off += step) {
            char* w = (char*)big + off;
            printf("  offset %6zu MiB:  as W -> O0 %.4f  O1 %.4f   | as out <- Wa %.4f  Wb %.4f\n", off >> 20, time_x(w, O[0], plane3, 2, 5),
                   time_x(w, O[1 % NO], plane3, 2, 5), time_x(Wfix[0], w, plane3, 2, 5), time_x(Wfix[1], w, plane3, 2, 5));
            fflush(stdout);
        }
        CK(hipFree(big));
        for (auto& w : Wfix) CK(hipFree(w));
    }
    return 0;
}
