// tools/membench4.hip -- developer measurement: which side of the Y -> X hand-over should be the scattered one, and how
// coarse must the scatter be?  Pure 2 GiB -> 2 GiB tile copies (512-thread workgroups, 8 x 16 B per lane, next tile's loads
// issued before the current tile's stores) with the access patterns a column pass can have on either side:
//   R/W "seg128@8K"  : 512 segments of 128 B, 8 KiB apart              (Y pass on the natural [x][y][z] layout)
//   R   "seg128@4M"  : 512 segments of 128 B, 4 MiB apart              (X pass reading [x][y][z])
//   R   "seg128@1K"  : 512 segments of 128 B, 1 KiB apart, 8 tiles share one contiguous 512 KiB region (blocked layout, G = 8)
//   W   "seg1K@512K" : 64 segments of 1 KiB, 512 KiB apart             (Y pass writing the blocked layout, G = 8)
//   R/W "run64K"     : one contiguous 64 KiB run                        (tile-major layout / staged transposed store)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(...)                                                                           \
    do {                                                                                  \
        hipError_t e_ = (__VA_ARGS__);                                                    \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

enum { SEG128_8K = 0, SEG128_4M = 1, SEG128_1K = 2, SEG1K_512K = 3, RUN64K = 4, SEG128_4M_PAD128 = 5, SEG128_4M_PAD1K = 6, SEG128_4M_PAD4K = 7, SEG128_4M_PAD64K = 8, SEG128_8K_PAD = 9, SEG128_ROWPAD_4M = 10, SEG128_ROWPAD_4M_PAD = 11, SEG128_4M_PAD384 = 12, SEG128_4M_PAD640 = 13, SEG128_1M = 14, SEG128_1M_ROT = 15, SEG128_4M_ROT3 = 16, SEG128_512K = 17, SEG128_512K_ROT = 18 };
static const char* kNames[] = {"seg128@8K", "seg128@4M", "seg128@1K", "seg1K@512K", "run64K", "seg128@4M+128", "seg128@4M+1K", "seg128@4M+4K", "seg128@4M+64K", "seg128@8K+128", "rowpad@4M+64K", "rowpad@4M+64K+128", "seg128@4M+384", "seg128@4M+640", "seg128@1M", "seg128@1M rot3", "seg128@4M rot3", "seg128@512K", "seg128@512K rot3"};

// element offset of point k (0..7) of thread tid in tile t, for a 512^3 volume of 16-byte elements
template <int PAT> __device__ __forceinline__ size_t addr(unsigned t, int tid, int k) {
    const int c = tid & 7, j = tid >> 3;       // column within the 128-byte line, row group
    const int idx = j + 64 * k;                // 0..511: position along the pass's FFT axis
    if (PAT == SEG128_8K) {                    // tile = (x, b): [x][idx][b*8 + c]
        const unsigned x = t >> 6, b = t & 63;
        return ((size_t)x * 512 + idx) * 512 + b * 8 + c;
    } else if (PAT == SEG128_4M) {             // tile = (y, b): [idx][y][b*8 + c]
        const unsigned y = t >> 6, b = t & 63;
        return ((size_t)idx * 512 + y) * 512 + b * 8 + c;
    } else if (PAT == SEG128_1M || PAT == SEG128_1M_ROT) {   // X pass at P = 4: [x][ys = 128][512], tile = (yy, b)
        const unsigned y = (t >> 6) & 127, b = t & 63;
        const unsigned bb = PAT == SEG128_1M_ROT ? (b + 3u * (unsigned)idx) & 63u : b;  // rows rotated by 3 lines per plane
        return ((size_t)idx * 128 + y) * 512 + bb * 8 + c;
    } else if (PAT == SEG128_512K || PAT == SEG128_512K_ROT) {   // P = 8: [x][ys = 64][512]
        const unsigned y = (t >> 6) & 63, b = t & 63;
        const unsigned bb = PAT == SEG128_512K_ROT ? (b + 3u * (unsigned)idx) & 63u : b;
        return ((size_t)idx * 64 + y) * 512 + bb * 8 + c;
    } else if (PAT == SEG128_4M_ROT3) {
        const unsigned y = t >> 6, b = t & 63;
        return ((size_t)idx * 512 + y) * 512 + ((b + 3u * (unsigned)idx) & 63u) * 8 + c;
    } else if (PAT == SEG128_8K_PAD) {         // Y pass on rows padded by one line: [x][idx][520]
        const unsigned x = t >> 6, b = t & 63;
        return ((size_t)x * 512 + idx) * 520 + b * 8 + c;
    } else if (PAT == SEG128_ROWPAD_4M || PAT == SEG128_ROWPAD_4M_PAD) {  // X pass on that layout (+ one line per plane)
        const unsigned y = t >> 6, b = t & 63;
        return (size_t)idx * (512 * 520 + (PAT == SEG128_ROWPAD_4M_PAD ? 8 : 0)) + (size_t)y * 520 + b * 8 + c;
    } else if (PAT >= SEG128_4M_PAD128) {      // as seg128@4M with the plane stride padded (not a power of two)
        const unsigned y = t >> 6, b = t & 63;
        const size_t pad = PAT == SEG128_4M_PAD128 ? 8 : (PAT == SEG128_4M_PAD1K ? 64 : (PAT == SEG128_4M_PAD4K ? 256 : (PAT == SEG128_4M_PAD384 ? 24 : (PAT == SEG128_4M_PAD640 ? 40 : 4096))));
        return (size_t)idx * (512 * 512 + pad) + (size_t)y * 512 + b * 8 + c;
    } else if (PAT == SEG128_1K) {             // tile = (g, b, r): [g][b][idx][r][c], G = 8
        const unsigned r = t & 7, gb = t >> 3;
        return (((size_t)gb * 512 + idx) * 8 + r) * 8 + c;
    } else if (PAT == SEG1K_512K) {            // tile = (x, b): [g = idx/8][b][x][idx%8][c]
        const unsigned x = t >> 6, b = t & 63;
        return ((((size_t)(idx >> 3) * 64 + b) * 512 + x) * 8 + (idx & 7)) * 8 + c;
    } else {                                   // one contiguous run: lane-interleaved
        return (size_t)t * 4096 + (size_t)k * 512 + tid;
    }
}

template <int RP, int WP, bool NTL, bool NTS> __global__ void __launch_bounds__(512) tile_copy(const d2* in, d2* out, unsigned ntiles) {
    const int tid = threadIdx.x;
    d2 v[8], w[8];
    auto load = [&](unsigned t, d2* d) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const d2* p = in + addr<RP>(t, tid, k);
            d[k] = NTL ? __builtin_nontemporal_load(p) : *p;
        }
    };
    if (blockIdx.x < ntiles) load(blockIdx.x, v);
    for (unsigned t = blockIdx.x; t < ntiles; t += gridDim.x) {
        if (t + gridDim.x < ntiles) load(t + gridDim.x, w);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            d2* p = out + addr<WP>(t, tid, k);
            if (NTS) __builtin_nontemporal_store(v[k], p);
            else *p = v[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = w[k];
    }
}

static hipStream_t g_s;
template <class F> static double time_ms(F&& f, int rounds = 7) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> ms;
    f();
    for (int r = 0; r < rounds; ++r) {
        CK(hipEventRecord(e0, g_s));
        f();
        CK(hipEventRecord(e1, g_s));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t);
    }
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

static d2 *g_a, *g_b;
template <int RP, int WP> void run() {
    const unsigned ntiles = 512u * 64u;
    for (int grid : {256, 512}) {
        double ms = time_ms([&] { hipLaunchKernelGGL((tile_copy<RP, WP, true, true>), dim3(grid), dim3(512), 0, g_s, g_a, g_b, ntiles); });
        double ms2 = time_ms([&] { hipLaunchKernelGGL((tile_copy<RP, WP, false, false>), dim3(grid), dim3(512), 0, g_s, g_a, g_b, ntiles); });
        printf("read %-11s write %-11s grid %4d   nt %7.3f ms %6.0f GB/s    plain %7.3f ms %6.0f GB/s\n", kNames[RP], kNames[WP], grid, ms,
               2.0 * 2147.483648 / ms, ms2, 2.0 * 2147.483648 / ms2);
        fflush(stdout);
    }
}

int main() {
    const size_t n = 512ull * 512 * 512;
    CK(hipMalloc(&g_a, n * 16 + 512ull * 8192 * 16));
    CK(hipMalloc(&g_b, n * 16 + 512ull * 8192 * 16));
    CK(hipMemset(g_a, 1, n * 16));
    CK(hipMemset(g_b, 0, n * 16));
    CK(hipStreamCreateWithFlags(&g_s, hipStreamNonBlocking));
    run<RUN64K, RUN64K>();
    run<RUN64K, SEG128_8K>();
    run<RUN64K, SEG1K_512K>();
    run<SEG128_8K, SEG128_8K>();
    run<SEG128_8K, RUN64K>();
    run<SEG128_8K, SEG1K_512K>();
    run<SEG128_4M, RUN64K>();
    run<SEG128_4M_PAD128, RUN64K>();
    run<SEG128_4M_PAD1K, RUN64K>();
    run<SEG128_4M_PAD4K, RUN64K>();
    run<SEG128_4M_PAD64K, RUN64K>();
    run<SEG128_1K, RUN64K>();
    run<SEG128_4M_PAD384, RUN64K>();
    run<SEG128_4M_PAD640, RUN64K>();
    run<SEG128_8K_PAD, SEG128_8K_PAD>();
    run<RUN64K, SEG128_8K_PAD>();
    run<SEG128_ROWPAD_4M, RUN64K>();
    run<SEG128_ROWPAD_4M_PAD, RUN64K>();
    run<SEG128_4M_ROT3, RUN64K>();
    run<SEG128_1M, RUN64K>();
    run<SEG128_1M_ROT, RUN64K>();
    run<SEG128_512K, RUN64K>();
    run<SEG128_512K_ROT, RUN64K>();
    return 0;
}
