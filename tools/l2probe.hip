// tools/l2probe.hip -- developer probe (not part of the library; round 5).  Two questions behind DESIGN section 4's ceiling argument:
//
//  A. What do the memory system's streams deliver TODAY (the figures DESIGN cited from round 2's removed membench harness):
//     read-only / write-only / copy with 16 B per lane from persistent workgroups, on buffers that fit the 256 MiB Infinity Cache
//     (128 MiB pair) and on 2 GiB buffers (HBM), every ordered pair of four 2 GiB allocations (placement classes of round 3).
//
//  B. Can the Z -> Y hand-over of the 2D stage stay inside ONE XCD's 4 MiB L2?  The one-launch stage (csrc/dfft_zy.hip) deals a
//     plane's row units to workgroups on all eight XCDs, so the intermediate crosses the fabric twice (sc1 write-through stores,
//     sc1 loads: PMC traffic 2.0 x algorithmic).  Here a "plane" of ROWS x COLS 16-byte elements is owned by ONE group of 32
//     workgroups: phase 1 copies whole rows in -> scratch (the Z pass's access pattern), the group synchronises on its own counter,
//     phase 2 reads 128-byte-wide column tiles of the scratch plane (the Y pass's pattern) and stores them.  Group = the 32
//     workgroups of one XCD (blockIdx % 8, checked against HW_REG_XCC_ID) or, as the control, 32 workgroups spread over all XCDs.
//     Knobs: phase-1 store flavour (plain keeps the line in L2, sc1 writes through and drops it), scratch = the full-size
//     hand-over buffer (in place / out of place) or a small ring of plane slots per group that stays hot in L2, one plane of
//     look-ahead or none.  Timed with HIP events; run under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` (argument `pmc`: one
//     launch per configuration, in the printed order) for the bytes that actually cross the L2's fabric side.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/l2probe.hip -o tools/bin/l2probe
//   l2probe [time|pmc] [GiB of data per buffer, default 2]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

typedef double   d2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

enum { AUX_PLAIN = 0, AUX_SC0 = 1, AUX_NT = 2, AUX_SC1 = 16 };
constexpr int THREADS = 512;
constexpr int NGROUPS = 8, GROUP_WGS = 32;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}
template <int AUX> __device__ __forceinline__ d2v bload(__amdgpu_buffer_rsrc_t rs, unsigned elem) {
    return __builtin_bit_cast(d2v, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(elem * 16u), 0, AUX));
}
template <int AUX> __device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t rs, unsigned elem, d2v v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), rs, (int)(elem * 16u), 0, AUX);
}

struct Ctl {
    unsigned arrive[NGROUPS * 64];  // one counter per group, 256 B apart
    unsigned xcc_mismatch;          // workgroups whose HW_REG_XCC_ID differs from blockIdx % 8
    unsigned timeouts;
};

// ---- part B: two phases per plane inside one group of 32 workgroups
// ST: flavour of the phase-1 stores into scratch; LD: flavour of the phase-2 loads from scratch
// OUTMODE 0: phase 2 stores in place on the scratch plane (plain stores; scratch must be full size), the result buffer is scratch
//         1: phase 2 stores to `out` with nt stores
template <int ST, int LD, int OUTMODE>
__global__ void __launch_bounds__(THREADS) two_phase_kernel(const d2v* in, d2v* out, d2v* scr, Ctl* ctl, int rows, int cols, int nplanes,
                                                            int spread, int slots, int ahead) {
    extern __shared__ char force_one_wg_per_cu[];
    __shared__ unsigned    sh_ok;
    const int              tid = threadIdx.x;
    int                    g, r;
    if (!spread) {
        g = blockIdx.x % NGROUPS;
        r = blockIdx.x / NGROUPS;
        const unsigned xcc = __builtin_amdgcn_s_getreg(6164) & 15u;  // HW_REG_XCC_ID (id 20), bits 3:0
        if (tid == 0 && xcc != (unsigned)g) atomicAdd(&ctl->xcc_mismatch, 1u);
    } else {
        g = blockIdx.x / GROUP_WGS;
        r = blockIdx.x % GROUP_WGS;
    }
    const size_t plane_elems = (size_t)rows * cols;
    const int    K = nplanes / NGROUPS;  // planes of this group: g + 8 k
    unsigned*    ctr = &ctl->arrive[g * 64];
    auto scratch_plane = [&](int k) -> d2v* {
        return slots > 0 ? scr + ((size_t)g * slots + (size_t)(k % slots)) * plane_elems : scr + ((size_t)g + (size_t)NGROUPS * k) * plane_elems;
    };
    auto phase1 = [&](int k) {  // whole rows: in -> scratch (8 rows per unit; a wave moves 1 KiB runs)
        const size_t               p = (size_t)g + (size_t)NGROUPS * k;
        const __amdgpu_buffer_rsrc_t ri = rsrc_of(in + p * plane_elems, plane_elems * 16), rs = rsrc_of(scratch_plane(k), plane_elems * 16);
        const int                  per = 8 * cols / THREADS;  // elements per thread per unit
        for (int u = r; u < rows / 8; u += GROUP_WGS) {
            d2v v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < per) v[i] = bload<AUX_NT>(ri, (unsigned)(u * 8 * cols + tid + THREADS * i));
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < per) bstore<ST>(rs, (unsigned)(u * 8 * cols + tid + THREADS * i), v[i]);
        }
    };
    auto phase2 = [&](int k) {  // column tiles of 8 elements (128 B) x rows: scratch -> result
        const size_t               p = (size_t)g + (size_t)NGROUPS * k;
        const __amdgpu_buffer_rsrc_t rs = rsrc_of(scratch_plane(k), plane_elems * 16), ro = rsrc_of(out + p * plane_elems, plane_elems * 16);
        const int                  per = rows * 8 / THREADS;
        const int                  c = tid % 8, j = tid / 8;
        for (int u = r; u < cols / 8; u += GROUP_WGS) {
            d2v v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < per) v[i] = bload<LD>(rs, (unsigned)((j + 64 * i) * cols + u * 8 + c));
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < per) {
                    if constexpr (OUTMODE == 0) bstore<AUX_PLAIN>(rs, (unsigned)((j + 64 * i) * cols + u * 8 + c), v[i]);
                    else bstore<AUX_NT>(ro, (unsigned)((j + 64 * i) * cols + u * 8 + c), v[i]);
                }
        }
    };
    auto arrive = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto wait = [&](int k) -> bool {
        if (tid == 0) {
            unsigned ok = 0;
            for (unsigned polls = 0; polls < 4000000u; ++polls) {
                if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)(GROUP_WGS * (k + 1))) {
                    ok = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (!ok) atomicAdd(&ctl->timeouts, 1u);
            sh_ok = ok;
        }
        __syncthreads();
        const bool ok = sh_ok != 0;
        __syncthreads();
        return ok;
    };
    if (ahead) {
        phase1(0);
        arrive();
        for (int k = 0; k < K; ++k) {
            if (k + 1 < K) {
                phase1(k + 1);
                arrive();
            }
            if (!wait(k)) return;
            phase2(k);
        }
    } else {
        for (int k = 0; k < K; ++k) {
            phase1(k);
            arrive();
            if (!wait(k)) return;
            phase2(k);
        }
    }
}

// ---- part A: plain streams.  mode 0 read-only (sum kept alive), 1 write-only, 2 copy; 16 B per lane, 8 per thread per tile
template <int MODE> __global__ void __launch_bounds__(THREADS) stream_kernel(const d2v* __restrict__ a, d2v* __restrict__ b, long long ntiles, d2v* sink) {
    extern __shared__ char force_one_wg_per_cu[];
    const int              tid = threadIdx.x;
    constexpr int          TILE = THREADS * 8;
    d2v                    acc = {0.0, 0.0};
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        d2v v[8];
        if constexpr (MODE != 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(a + t * TILE + k * THREADS + tid);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = d2v{(double)t, (double)k};
        }
        if constexpr (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += v[k];
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(v[k], b + t * TILE + k * THREADS + tid);
        }
    }
    if (MODE == 0 && acc.x == 1.2345e300) sink[0] = acc;
}

__global__ void fill_kernel(d2v* a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = d2v{(double)i, (double)(i ^ 0x5a5a)};
}
__global__ void cmp_kernel(const d2v* a, const d2v* b, size_t n, unsigned long long* bad) {
    unsigned long long m = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const d2v x = a[i], y = b[i];
        m += (x.x != y.x) || (x.y != y.y);
    }
    if (m) atomicAdd(bad, m);
}

static hipStream_t g_s;
static float time_ms(const std::function<void()>& f, int reps, float* best) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> t;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0, g_s));
        f();
        CK(hipEventRecord(e1, g_s));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    *best = t.front();
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return t[t.size() / 2];
}

typedef void (*two_phase_fn)(const d2v*, d2v*, d2v*, Ctl*, int, int, int, int, int, int);
struct StoreFlavour {
    const char*  name;
    two_phase_fn inplace, outofplace;
};

int main(int argc, char** argv) {
    const bool   pmc = argc > 1 && !strcmp(argv[1], "pmc");
    const double gib = argc > 2 ? atof(argv[2]) : 2.0;
    const size_t n = (size_t)(gib * (1ull << 30)) / 16;  // elements per buffer
    const int    reps = pmc ? 1 : 7;
    CK(hipStreamCreateWithFlags(&g_s, hipStreamNonBlocking));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# l2probe: %s, %d CUs, %.2f GiB per buffer, mode %s\n", prop.name, cus, gib, pmc ? "pmc (one launch per line)" : "time");
    if (cus != NGROUPS * GROUP_WGS) {
        printf("expected 256 CUs\n");
        return 1;
    }
    const size_t LDS_FORCE = 96 * 1024;  // one workgroup per CU
    d2v*         buf[4];
    for (int i = 0; i < 4; ++i) CK(hipMalloc(&buf[i], n * 16));
    Ctl* ctl;
    CK(hipMalloc(&ctl, sizeof(Ctl)));
    unsigned long long* bad;
    CK(hipMalloc(&bad, 8));
    d2v* ring;
    CK(hipMalloc(&ring, (size_t)NGROUPS * 4 * (4u << 20)));  // up to 4 slots of 4 MiB per group
    fill_kernel<<<2048, 256, 0, g_s>>>(buf[0], n);
    CK(hipStreamSynchronize(g_s));

    // ---------------- part A
    {
        printf("\n## A. streams (persistent, one 512-thread workgroup per CU, 16 B per lane, nt)\n");
        printf("%-44s %10s %10s %10s\n", "case", "median ms", "best ms", "TB/s(med)");
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_FORCE));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_FORCE));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_FORCE));
        auto run = [&](const char* name, int mode, const d2v* a, d2v* b, size_t elems, int inner) {
            const long long ntiles = (long long)(elems / (THREADS * 8));
            auto            f = [&]() {
                for (int i = 0; i < inner; ++i) {
                    if (mode == 0) stream_kernel<0><<<cus, THREADS, LDS_FORCE, g_s>>>(a, b, ntiles, buf[3]);
                    if (mode == 1) stream_kernel<1><<<cus, THREADS, LDS_FORCE, g_s>>>(a, b, ntiles, buf[3]);
                    if (mode == 2) stream_kernel<2><<<cus, THREADS, LDS_FORCE, g_s>>>(a, b, ntiles, buf[3]);
                }
            };
            f();  // warm (and, for the small buffers, make them cache resident)
            float       best, med = time_ms(f, reps, &best);
            const double bytes = (double)elems * 16 * (mode == 2 ? 2 : 1) * inner;
            printf("%-44s %10.4f %10.4f %10.3f\n", name, med / inner, best / inner, bytes / (med * 1e-3) / 1e12);
        };
        const size_t small = (64ull << 20) / 16;  // 64 MiB per side: a 128 MiB pair inside the 256 MiB Infinity Cache
        run("read  64 MiB (cache resident)", 0, buf[0], buf[1], small, 8);
        run("write 64 MiB (cache resident)", 1, buf[0], buf[1], small, 8);
        run("copy  64 MiB -> 64 MiB (cache resident)", 2, buf[0], buf[1], small, 8);
        run("read  2 GiB (HBM)", 0, buf[0], buf[1], n, 1);
        run("write 2 GiB (HBM)", 1, buf[0], buf[1], n, 1);
        if (!pmc) {
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    if (i == j) continue;
                    char nm[64];
                    snprintf(nm, sizeof nm, "copy  2 GiB buf%d -> buf%d (HBM)", i, j);
                    run(nm, 2, buf[i], buf[j], n, 1);
                }
        } else {
            run("copy  2 GiB buf0 -> buf1 (HBM)", 2, buf[0], buf[1], n, 1);
        }
        fill_kernel<<<2048, 256, 0, g_s>>>(buf[0], n);
        CK(hipStreamSynchronize(g_s));
    }

    // ---------------- part B
    printf("\n## B. two phases per plane inside a group of 32 workgroups (rows in -> scratch, group sync, 128-byte column tiles scratch -> result)\n");
    printf("# bytes moved by the algorithm per launch: %.3f GB (read in + write result); the hand-over adds the same again if it crosses the fabric\n",
           2.0 * n * 16 / 1e9);
    printf("%-3s %-9s %-7s %-6s %-10s %-5s %10s %10s %9s %8s %5s\n", "#", "plane", "group", "store", "scratch", "ahead", "median ms", "best ms", "TB/s alg", "bad", "tmo");
    const StoreFlavour fl[3] = {
        {"plain", two_phase_kernel<AUX_PLAIN, AUX_SC1, 0>, two_phase_kernel<AUX_PLAIN, AUX_SC1, 1>},
        {"sc1", two_phase_kernel<AUX_SC1, AUX_SC1, 0>, two_phase_kernel<AUX_SC1, AUX_SC1, 1>},
        {"nt", two_phase_kernel<AUX_NT, AUX_SC1, 0>, two_phase_kernel<AUX_NT, AUX_SC1, 1>},
    };
    for (auto& f : fl) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(f.inplace), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_FORCE));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(f.outofplace), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_FORCE));
    }
    struct Shape {
        int rows, cols;
    } shapes[] = {{512, 512}, {256, 512}, {128, 512}, {256, 256}};
    int line = 0;
    for (auto sh : shapes) {
        const size_t pe = (size_t)sh.rows * sh.cols;
        const int    nplanes = (int)(n / pe) / NGROUPS * NGROUPS;
        for (int spread = 0; spread < 2; ++spread)
            for (int st = 0; st < 3; ++st) {
                if (st == 2 && spread) continue;
                // scratch: 0 = full size in place, 1 = full size out of place, 2 = ring of 3 slots out of place, 3 = ring of 2 slots (no look-ahead only)
                for (int sc = 0; sc < 4; ++sc)
                    for (int ahead = 0; ahead < 2; ++ahead) {
                        if (sc == 3 && ahead) continue;
                        if (spread && (sc == 3)) continue;
                        const int    slots = sc == 2 ? 3 : sc == 3 ? 2 : 0;
                        two_phase_fn k = sc == 0 ? fl[st].inplace : fl[st].outofplace;
                        d2v*         scr = sc >= 2 ? ring : buf[1];
                        d2v*         res = sc == 0 ? buf[1] : buf[2];
                        auto         f = [&]() {
                            CK(hipMemsetAsync(ctl, 0, sizeof(Ctl), g_s));
                            k<<<cus, THREADS, LDS_FORCE, g_s>>>(buf[0], buf[2], scr, ctl, sh.rows, sh.cols, nplanes, spread, slots, ahead);
                        };
                        if (!pmc) f();
                        float best, med = time_ms(f, reps, &best);
                        CK(hipMemsetAsync(bad, 0, 8, g_s));
                        cmp_kernel<<<2048, 256, 0, g_s>>>(buf[0], res, (size_t)nplanes * pe, bad);
                        unsigned long long hb = 0;
                        Ctl                hc;
                        CK(hipMemcpyAsync(&hb, bad, 8, hipMemcpyDeviceToHost, g_s));
                        CK(hipMemcpyAsync(&hc, ctl, sizeof(Ctl), hipMemcpyDeviceToHost, g_s));
                        CK(hipStreamSynchronize(g_s));
                        char pl[16], scn[16];
                        snprintf(pl, sizeof pl, "%dx%d", sh.rows, sh.cols);
                        snprintf(scn, sizeof scn, "%s", sc == 0 ? "full-inpl" : sc == 1 ? "full-oop" : sc == 2 ? "ring3" : "ring2");
                        printf("%-3d %-9s %-7s %-6s %-10s %-5d %10.4f %10.4f %9.3f %8llu %5u%s\n", line++, pl, spread ? "spread" : "xcd", fl[st].name, scn, ahead, med, best,
                               2.0 * nplanes * pe * 16 / (med * 1e-3) / 1e12, hb, hc.timeouts, hc.xcc_mismatch ? "  XCC!=blockIdx%8" : "");
                        fflush(stdout);
                    }
            }
    }
    return 0;
}
