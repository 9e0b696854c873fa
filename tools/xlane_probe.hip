// xlane_probe.hip -- what does the wave-local exchange of the 4-stage column plans cost in LDS, and what would it cost across lanes?
// (VERDICT r05 Weak-5 / Next-1c, round 6: "a cross-lane form of the one radix stage whose partners sit in one wavefront".)
//
// The exchange in question (csrc/dfft_fft_impl.h, run_stages, the wave-owned exchange after the stage R = 8, NS = 8 of 1024 = 8 8 8 2 on
// 16 points x 64 butterfly threads, 8-column tiles of 16-byte elements): under the wave-interleaved labelling a wavefront holds the
// butterfly ids w + 8 g (g = lane / 8, column c = lane % 8), and the scatter / gather between the two radix-8 stages is, for each of a
// thread's two butterflies, an 8 x 8 TRANSPOSE between g (lane bits 3..5) and the register index r -- 16 bytes per element.
//
//   mode 0  LDS: 8 ds_write_b128 + wave-local wait + 8 ds_read_b128 per butterfly, the library's positions (lds_pos swizzle)
//   mode 1  gfx950 cross-lane: three butterfly rounds -- lane bit 5 by v_permlane32_swap, bit 4 by v_permlane16_swap, bit 3 by
//           v_mov_b32_dpp row_ror:8 with bank masks -- no LDS, no address registers
//   mode 2  generic shuffles: the same three rounds through ds_bpermute_b32 (__shfl_xor) + selects
//   mode 3  no exchange (loop and arithmetic only)
// Every mode also does FMAS dependent fma per element and iteration (the arithmetic the exchange hides behind; 0 = exchange alone).
//
// usage: xlane_probe [iterations, default 4000]     prints ns and CU clocks per two-butterfly exchange for 1 / 2 workgroups per CU
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/xlane_probe.hip -o tools/bin/xlane_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(stmt)                                                                                 \
    do {                                                                                            \
        hipError_t e_ = (stmt);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            fprintf(stderr, "[%s:%d] %s: %s\n", __FILE__, __LINE__, #stmt, hipGetErrorString(e_)); \
            return 2;                                                                               \
        }                                                                                           \
    } while (0)

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int THREADS = 512, R = 8, Q = 2;  // 8 waves, radix 8, two butterflies per thread (16 points)

// one round of the transposition on a register pair: lanes with the lane bit clear keep a and receive the partner's a into b, lanes with
// the bit set keep b and receive the partner's b into a
template <int BIT> __device__ __forceinline__ void round_native(u32x4& a, u32x4& b) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if constexpr (BIT == 5) {  // upper half of a <-> lower half of b
            u32x2 r = __builtin_amdgcn_permlane32_swap(a[d], b[d], false, false);
            a[d] = r.x;
            b[d] = r.y;
        } else if constexpr (BIT == 4) {  // odd rows of a <-> even rows of b
            u32x2 r = __builtin_amdgcn_permlane16_swap(a[d], b[d], false, false);
            a[d] = r.x;
            b[d] = r.y;
        } else {  // lanes 8..15 of every row of a <- lanes 0..7 of b, lanes 0..7 of b <- lanes 8..15 of a
            const unsigned na = __builtin_amdgcn_update_dpp(a[d], b[d], 0x128 /* row_ror:8 */, 0xf, 0xc, false);
            const unsigned nb = __builtin_amdgcn_update_dpp(b[d], a[d], 0x128, 0xf, 0x3, false);
            a[d] = na;
            b[d] = nb;
        }
    }
}
template <int BIT> __device__ __forceinline__ void round_shfl(u32x4& a, u32x4& b, int lane) {
    const bool hi = (lane >> BIT) & 1;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const unsigned send = hi ? a[d] : b[d];
        const unsigned recv = __shfl_xor(send, 1 << BIT, 64);
        if (hi) a[d] = recv;
        else b[d] = recv;
    }
}

template <int MODE, int FMAS> __global__ __launch_bounds__(THREADS) void probe(u32x4* __restrict__ out, const u32x4* __restrict__ in, int iters, double w) {
    __shared__ u32x4 lds[THREADS * R];  // 64 KiB: one butterfly of every thread at a time
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 3, c = lane & 7;
    u32x4       v[Q][R];
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) v[q][r] = in[((size_t)blockIdx.x * THREADS + tid) * (Q * R) + q * R + r];
    u32x4* mine = lds + wave * (64 * R);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            if constexpr (MODE == 0) {
                // value (g, r) -> position r * 8 + g, read back by lane group g' = r as its register g; the 128-byte halves of a
                // 256-byte bank row swapped in every second group of 8 positions (lds_pos of the library)
#pragma unroll
                for (int r = 0; r < R; ++r) mine[((r * 8 + g) ^ (r & 1)) * 8 + c] = v[q][r];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < R; ++r) v[q][r] = mine[((g * 8 + r) ^ (g & 1)) * 8 + c];
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            } else if constexpr (MODE == 1) {
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    if (!(i & 4)) round_native<5>(v[q][i], v[q][i | 4]);
                }
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    if (!(i & 2)) round_native<4>(v[q][i], v[q][i | 2]);
                }
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    if (!(i & 1)) round_native<3>(v[q][i], v[q][i | 1]);
                }
            } else if constexpr (MODE == 2) {
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    if (!(i & 4)) round_shfl<5>(v[q][i], v[q][i | 4], lane);
                }
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    if (!(i & 2)) round_shfl<4>(v[q][i], v[q][i | 2], lane);
                }
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    if (!(i & 1)) round_shfl<3>(v[q][i], v[q][i | 1], lane);
                }
            }
            // the arithmetic next to it: FMAS dependent fp64 fma per 16-byte element (a radix-8 butterfly + twiddles is ~ 12 per element)
            if constexpr (FMAS > 0) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    double x = __hiloint2double((int)v[q][r][1], (int)v[q][r][0]), y = __hiloint2double((int)v[q][r][3], (int)v[q][r][2]);
#pragma unroll
                    for (int f = 0; f < FMAS; f += 2) {
                        x = __builtin_fma(y, w, x);
                        y = __builtin_fma(x, -w, y);
                    }
                    v[q][r][0] = (unsigned)__double2loint(x);
                    v[q][r][1] = (unsigned)__double2hiint(x);
                    v[q][r][2] = (unsigned)__double2loint(y);
                    v[q][r][3] = (unsigned)__double2hiint(y);
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) out[((size_t)blockIdx.x * THREADS + tid) * (Q * R) + q * R + r] = v[q][r];
}

template <int MODE, int FMAS> static int run(const char* name, int iters, u32x4* din, u32x4* dout, int max_blocks, std::vector<u32x4>& host, bool check, double clock_ghz, int cus) {
    // correctness: ONE exchange of tagged elements (w = 0 leaves them alone when FMAS = 0; checked only with FMAS = 0)
    if (check) {
        probe<MODE, FMAS><<<dim3(2), dim3(THREADS), 0, 0>>>(dout, din, 1, 0.0);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(host.data(), dout, sizeof(u32x4) * 2 * THREADS * Q * R, hipMemcpyDeviceToHost));
        long bad = 0;
        for (int b = 0; b < 2; ++b)
            for (int t = 0; t < THREADS; ++t)
                for (int q = 0; q < Q; ++q)
                    for (int r = 0; r < R; ++r) {
                        const int lane = t & 63, g = lane >> 3;
                        // register r of lane group g must now hold what lane group r held in register g (MODE 3: nothing moved)
                        const int st = MODE == 3 ? t : (t & ~56) | (r << 3), sr = MODE == 3 ? r : g;
                        const unsigned want = ((unsigned)b << 24) | ((unsigned)st << 8) | (unsigned)(q * R + sr);
                        const u32x4 got = host[((size_t)b * THREADS + t) * (Q * R) + q * R + r];
                        if (got[0] != want || got[1] != ~want || got[2] != want + 1 || got[3] != want * 3u) ++bad;
                    }
        printf("%-34s check: %ld wrong elements of %d\n", name, bad, 2 * THREADS * Q * R);
        if (bad) return 1;
    }
    for (int per_cu = 1; per_cu <= 2; ++per_cu) {
        const int blocks = cus * per_cu;
        if (blocks > max_blocks) break;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        probe<MODE, FMAS><<<dim3(blocks), dim3(THREADS), 0, 0>>>(dout, din, 64, 1e-9);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            probe<MODE, FMAS><<<dim3(blocks), dim3(THREADS), 0, 0>>>(dout, din, iters, 1e-9);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double ns = best * 1e6 / iters;  // per iteration = one two-butterfly exchange of every thread of `per_cu` workgroups on a CU
        printf("%-34s fma/elem %2d  %d workgroup(s) per CU: %8.1f ns per tile exchange  = %7.0f CU clocks  (%5.1f clocks per wavefront and butterfly)\n", name, FMAS, per_cu,
               ns / per_cu, ns * clock_ghz / per_cu, ns * clock_ghz / (per_cu * 8 * Q));
        CHECK(hipEventDestroy(e0));
        CHECK(hipEventDestroy(e1));
    }
    return 0;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int    cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    printf("# %s, %d CUs, %.2f GHz; 512-thread workgroups, 16 points of 16 bytes per thread, one iteration = the 8 x 8 transposes of both butterflies\n", prop.name, cus, ghz);
    const int    max_blocks = 2 * cus;
    const size_t n = (size_t)max_blocks * THREADS * Q * R;
    std::vector<u32x4> host(n);
    for (size_t b = 0; b < (size_t)max_blocks; ++b)
        for (int t = 0; t < THREADS; ++t)
            for (int k = 0; k < Q * R; ++k) {
                const unsigned tag = ((unsigned)(b & 255) << 24) | ((unsigned)t << 8) | (unsigned)k;
                host[(b * THREADS + t) * (Q * R) + k] = u32x4{tag, ~tag, tag + 1, tag * 3u};
            }
    u32x4 *din, *dout;
    CHECK(hipMalloc(&din, n * sizeof(u32x4)));
    CHECK(hipMalloc(&dout, n * sizeof(u32x4)));
    CHECK(hipMemcpy(din, host.data(), n * sizeof(u32x4), hipMemcpyHostToDevice));
    int rc = 0;
    rc |= run<3, 0>("3 no exchange", iters, din, dout, max_blocks, host, true, ghz, cus);
    rc |= run<0, 0>("0 LDS, wave-local", iters, din, dout, max_blocks, host, true, ghz, cus);
    rc |= run<1, 0>("1 permlane32/16_swap + dpp ror:8", iters, din, dout, max_blocks, host, true, ghz, cus);
    rc |= run<2, 0>("2 ds_bpermute shuffles", iters, din, dout, max_blocks, host, true, ghz, cus);
    rc |= run<3, 12>("3 no exchange", iters, din, dout, max_blocks, host, false, ghz, cus);
    rc |= run<0, 12>("0 LDS, wave-local", iters, din, dout, max_blocks, host, false, ghz, cus);
    rc |= run<1, 12>("1 permlane32/16_swap + dpp ror:8", iters, din, dout, max_blocks, host, false, ghz, cus);
    rc |= run<2, 12>("2 ds_bpermute shuffles", iters, din, dout, max_blocks, host, false, ghz, cus);
    return rc;
}
