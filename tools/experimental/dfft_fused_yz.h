// dfft_fused_yz.h -- t0 as ONE persistent kernel: the 2D YZ FFT of a plane done by the workgroups of one XCD, with the
// Z->Y intermediate handed over through that XCD's L2 instead of a slab-sized buffer in HBM / Infinity Cache.
//
// Reference stage being replaced (behaviour): fftZY, /root/reference/3dmpifft_opt/include/fft_mpi_3d_api.cpp:466-522
// (per-plane batched Z FFT, then batched Y FFT), plus the t1 pack (kernel_func.cpp:73-86) through the store map.
//
// Why: the two-launch t0 (Z rows over a cache-sized chunk of planes, then Y columns over the same chunk) moves every
// element across the XCD <-> memory fabric four times (Z read, Z write, Y read, Y write).  A plane's 2D FFT contains a
// full transpose, so it cannot live in one CU's 160 KiB LDS -- but the 32 CUs of an XCD share a 4 MiB L2.  Here a *team*
// (all resident workgroups of one XCD) owns a plane at a time:
//     rows   : every workgroup Z-transforms its share of the plane's rows (one row per wavefront, no s_barrier) and
//              stores them into the team's scratch plane S (plain stores: written through L1, kept in the XCD's L2);
//     barrier A (team-wide, XCD-local counters);
//     columns: every workgroup loads its 128-byte-wide column tiles of S with L1-bypassing loads (served by the same
//              L2), arrives at barrier B as soon as the data sits in registers, Y-transforms and stores the result
//              through the pass's address map (natural layout or the packed send layout).
// HBM / fabric traffic of t0 drops from 4*S to 2*S bytes per element when S stays L2-resident.
//
// Correctness does not depend on dispatch order or on which XCD a workgroup lands on: teams are formed at run time from
// the hardware XCC_ID of each workgroup, so all members of a team share one physical L2 by construction; producer
// stores are drained (s_waitcnt vmcnt(0)) before the arrival atomic, consumers poll one word and then read S with
// loads that bypass the (never refreshed) per-CU L1.  All spins are bounded (wall clock); a time-out sets ctl->error,
// every workgroup leaves, and the host falls back to the two-launch path.
#pragma once
#include "dfft_fft_impl.h"

namespace dfft {

struct alignas(128) FusedTeam {
    unsigned cntA;
    unsigned padA[31];
    unsigned genA;
    unsigned padB[31];
    unsigned cntB;
    unsigned padC[31];
    unsigned genB;
    unsigned padD[31];
};
// zeroed (hipMemsetAsync) before every launch
struct alignas(128) FusedCtl {
    unsigned  registered;  // workgroups that have registered with their XCC
    unsigned  pad0[31];
    unsigned  error;       // != 0: a bounded spin gave up; everybody leaves
    unsigned  pad1[31];
    unsigned  xcc_count[16];
    unsigned  pad2[16];
    FusedTeam team[32];    // [xcc * TEAMS + sub]
};

enum { FUSED_ERR_REGISTER = 1, FUSED_ERR_WAIT_A = 2, FUSED_ERR_WAIT_B = 3, FUSED_ERR_TEAM_SIZE = 4 };

struct FusedCfgDefault {
    static constexpr int THREADS = 512;
    static constexpr int TEAMS = 1;     // independent teams per XCD (workgroup local index % TEAMS)
    static constexpr bool SHARE_S = false;  // TEAMS == 2 only: the two teams of an XCD take turns on ONE scratch plane --
                                            // while one team exchanges through the L2, the other streams HBM
    static constexpr bool IN_NT = true, OUT_NT = true;  // cache policy of the HBM side
    static constexpr int SLOAD = 0;     // how S is read: 0 = nt loads, 1 = buffer loads with sc1 (both bypass L1)
    static constexpr bool SSTORE_NT = false;
    static constexpr bool PREFETCH = true;   // load the next plane's first row unit before waiting at barrier A
    static constexpr bool MATH = true;       // false: data movement only (measurement builds)
    static constexpr int SPAD = 8;           // elements of padding per scratch row (spreads column reads over L2 channels)
    static constexpr int MIN_WAVES = 0;      // amdgpu_waves_per_eu lower bound
    static constexpr bool TW_RELOAD = false; // fetch each phase's twiddle set (L1/L2-resident table) when the phase starts
                                             // instead of keeping both sets in VGPRs for the whole kernel
    static constexpr unsigned TIMEOUT_TICKS = 20u * 1000u * 100u;  // 20 ms of the 100 MHz wall clock
};

#define DFFT_AGENT __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ unsigned fused_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, DFFT_AGENT); }

// one thread: wait until *gen >= it (or somebody reported an error / the wall clock runs out)
template <unsigned LIMIT> __device__ __forceinline__ bool fused_wait(const unsigned* gen, unsigned it, FusedCtl* ctl, unsigned code) {
    if (fused_ld(gen) >= it) return true;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        __builtin_amdgcn_s_sleep(1);
        if (fused_ld(gen) >= it) return true;
        if (fused_ld(&ctl->error) != 0) return false;
        if (wall_clock64() - t0 > LIMIT) {
            __hip_atomic_store(&ctl->error, code, __ATOMIC_RELAXED, DFFT_AGENT);
            return false;
        }
    }
}
// one thread: arrive at instance `it` of a team barrier of `T` members; the last arriver publishes the generation
__device__ __forceinline__ void fused_arrive(unsigned* cnt, unsigned* gen, unsigned it, unsigned T) {
    const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, DFFT_AGENT);
    if (old + 1u == it * T) __hip_atomic_store(gen, it, __ATOMIC_RELAXED, DFFT_AGENT);
}

template <class V> struct FusedScratch {
    const V*               base;
    __amdgpu_buffer_rsrc_t rsrc;
};

template <int MODE, class V> __device__ __forceinline__ V fused_sload(const FusedScratch<V>& s, unsigned elem) {
    if constexpr (MODE == 1) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        static_assert(sizeof(V) == 16, "buffer path: 16-byte elements");
        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(s.rsrc, (int)(elem * 16u), 0, 16 /* sc1 */);
        return __builtin_bit_cast(V, r);
    } else {
        return gload<true>(s.base + elem);
    }
}

// in  : [plane][N1][N2] (natural slab layout), plane stride in_plane_stride elements
// out : plane a, tile b (CB columns of z), FFT index idx (= ky), column c:
//       out + a*otile.a_stride + b*CB*otile.b_stride + map(omap, idx) + c*omap.cstride      (as fft_tiles_kernel)
// scratch: one (N1 x (N2 + SPAD)) plane per team
template <class V, class PZ, class PY, int DIR, class Cfg>
__global__ void __attribute__((amdgpu_flat_work_group_size(Cfg::THREADS, Cfg::THREADS),
                               amdgpu_waves_per_eu(Cfg::MIN_WAVES > 0 ? Cfg::MIN_WAVES : 1)))
fused_yz_kernel(const V* in, V* out, V* scratch, FusedCtl* ctl, const V* __restrict__ twz, const V* __restrict__ twy,
                AxisMap omap, TileMap otile, long long in_plane_stride, unsigned nplanes, unsigned a_first) {
    constexpr int THREADS = Cfg::THREADS;
    constexpr int N2 = PZ::N, N1 = PY::N;
    constexpr int EZ = PZ::E, TZ = PZ::T, EY = PY::E, TY = PY::T;
    static_assert(THREADS % TZ == 0 && TZ <= 64 && 64 % TZ == 0, "rows: one FFT inside one wavefront");
    constexpr int GR = THREADS / TZ;  // rows per row unit
    static_assert(THREADS % TY == 0, "columns: the workgroup is one tile");
    constexpr int CB = THREADS / TY;  // columns per tile
    static_assert(CB * sizeof(V) >= 64, "column tiles narrower than half a cache line");
    static_assert(N1 % GR == 0 && N2 % CB == 0, "plane must split into whole units");
    constexpr int UR = N1 / GR, UC = N2 / CB;
    constexpr int PITCH = N2 + Cfg::SPAD;
    constexpr int EMAX = EZ > EY ? EZ : EY;
    constexpr bool TWPOW = true;
    constexpr int  ROW_LDS = N2 + N2 / 8;  // padded row (lds_index<1, true>)
    constexpr unsigned LIMIT = Cfg::TIMEOUT_TICKS;

    extern __shared__ __attribute__((aligned(16))) char dfft_smem[];
    unsigned* shw = reinterpret_cast<unsigned*>(dfft_smem);  // [0] team, [1] member, [2] T, [3] NT, [4] rank, [5] ok
    V*        lds = reinterpret_cast<V*>(dfft_smem + 64);

    const int tid = threadIdx.x;
    // ---- registration: which XCD am I on, who else is ----
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 15u;
        const unsigned local = __hip_atomic_fetch_add(&ctl->xcc_count[xcc], 1u, __ATOMIC_RELAXED, DFFT_AGENT);
        const unsigned done = __hip_atomic_fetch_add(&ctl->registered, 1u, __ATOMIC_RELAXED, DFFT_AGENT) + 1u;
        bool ok = true;
        if (done != gridDim.x) ok = fused_wait<LIMIT>(&ctl->registered, gridDim.x, ctl, FUSED_ERR_REGISTER);
        unsigned nt = 0, rank = 0, T = 0, nx = 0, xrank = 0, Tp = 0;
        const unsigned sub = local % Cfg::TEAMS;
        for (unsigned x = 0; x < 16; ++x) {
            const unsigned n = fused_ld(&ctl->xcc_count[x]);
            if (n > 0) {
                if (x == xcc) xrank = nx;
                ++nx;
            }
            for (unsigned s = 0; s < (unsigned)Cfg::TEAMS; ++s) {
                const unsigned size = n > s ? (n - s + Cfg::TEAMS - 1) / Cfg::TEAMS : 0u;
                if (size == 0) continue;
                if (x == xcc && s == sub) {
                    rank = nt;
                    T = size;
                }
                if (x == xcc && s != sub) Tp = size;
                ++nt;
            }
        }
        shw[0] = xcc * Cfg::TEAMS + sub;
        shw[1] = local / Cfg::TEAMS;
        shw[2] = T;
        shw[3] = nt;
        shw[4] = rank;
        shw[5] = ok ? 1u : 0u;
        shw[6] = nx;
        shw[7] = xrank;
        shw[8] = Tp;
        shw[9] = xcc;
    }
    __syncthreads();
    if (shw[5] == 0) return;
    static_assert(!Cfg::SHARE_S || Cfg::TEAMS == 2, "a scratch plane is shared by exactly two teams");
    const unsigned team = shw[0], member = shw[1], T = shw[2];
    const unsigned sub = team % Cfg::TEAMS;
    const bool     paired = Cfg::SHARE_S && shw[8] > 0;  // the XCD's other team exists
    // first plane / plane step of this team.  SHARE_S: the XCD walks its planes in order, the teams alternate
    const unsigned rank = Cfg::SHARE_S ? shw[7] + shw[6] * (paired ? sub : 0u) : shw[4];
    const unsigned NT = Cfg::SHARE_S ? shw[6] * (paired ? 2u : 1u) : shw[3];
    FusedTeam*     tm = &ctl->team[team];
    FusedTeam*     tp = &ctl->team[team ^ 1u];  // SHARE_S: the partner team
    V*             S = scratch + (size_t)(Cfg::SHARE_S ? shw[9] : team) * N1 * PITCH;
    FusedScratch<V> Sr;
    Sr.base = S;
    Sr.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)S, 0, (int)((size_t)N1 * PITCH * sizeof(V)), 0x00020000);

    // ---- per-thread geometry ----
    const int gz = tid / TZ, jz = tid % TZ;  // row phase: row gz of the unit, butterfly id jz
    const int cy = tid % CB, jy = tid / CB;  // column phase
    V*        lds_row = lds + gz * ROW_LDS;

    constexpr int TWNZ = TwTotal<PZ, TWPOW>::value, TWNY = TwTotal<PY, TWPOW>::value;
    constexpr bool KEEP_TW = Cfg::MATH && !Cfg::TW_RELOAD;
    V twzr[KEEP_TW && TWNZ > 0 ? TWNZ : 1], twyr[KEEP_TW && TWNY > 0 ? TWNY : 1];
    if constexpr (KEEP_TW) {
        load_twiddles<V, PZ, 0, DIR, TWPOW>(twzr, twz, jz);
        load_twiddles<V, PY, 0, DIR, TWPOW>(twyr, twy, jy);
    }
    unsigned orel[EY];
#pragma unroll
    for (int k = 0; k < EY; ++k) {
        const int idx = jy + TY * k;
        const int ob = idx / omap.blk;
        orel[k] = (unsigned)(block_term(omap, ob) + (idx - ob * omap.blk) * omap.stride + cy * omap.cstride);
    }

    V v[EMAX], w[Cfg::PREFETCH ? EZ : 1];
    auto load_rows = [&](unsigned p, unsigned u, V* dst) {
        const V* ip = in + (long long)(a_first + p) * in_plane_stride + (long long)(u * GR + gz) * N2 + jz;
#pragma unroll
        for (int k = 0; k < EZ; ++k) dst[k] = gload<Cfg::IN_NT>(ip + TZ * k);
    };
    if constexpr (Cfg::PREFETCH) {
        if (rank < nplanes && member < (unsigned)UR) load_rows(rank, member, w);
    }
    unsigned it = 0;
    for (unsigned p = rank; p < nplanes; p += NT) {
        ++it;
        __syncthreads();  // LDS: the previous plane's column exchange is finished before rows scatter again
        // ---- rows of plane p ----
        bool first = true;
        for (unsigned u = member; u < (unsigned)UR; u += T) {
            if (Cfg::PREFETCH && first) {
#pragma unroll
                for (int k = 0; k < EZ; ++k) v[k] = w[k];
            } else {
                load_rows(p, u, v);
            }
            if constexpr (Cfg::MATH) {
                if constexpr (Cfg::TW_RELOAD) {
                    V twl[TWNZ > 0 ? TWNZ : 1];
                    load_twiddles<V, PZ, 0, DIR, TWPOW>(twl, twz, jz);
                    run_stages<V, PZ, 0, DIR, 1, true, true, TW_REG, TWPOW>(v, twl, lds_row, jz, 0);
                } else {
                    run_stages<V, PZ, 0, DIR, 1, true, true, TW_REG, TWPOW>(v, twzr, lds_row, jz, 0);
                }
            }
            if (first) {
                first = false;
                // S is free once every member of the team that used it last has its columns in registers: this team's
                // previous plane, or (SHARE_S) the partner's plane just before this one in the XCD's order
                const unsigned* gb = paired ? &tp->genB : &tm->genB;
                const unsigned  need = paired && sub == 1 ? it : it - 1;
                if (need > 0) {
                    if (tid == 0) shw[5] = fused_wait<LIMIT>(gb, need, ctl, FUSED_ERR_WAIT_B) ? 1u : 0u;
                    __syncthreads();
                    if (shw[5] == 0) return;
                }
            }
            V* sp = S + (size_t)(u * GR + gz) * PITCH + jz;
#pragma unroll
            for (int k = 0; k < EZ; ++k) gstore<Cfg::SSTORE_NT>(sp + TZ * k, v[k]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my rows have reached the L2
        __syncthreads();
        if (tid == 0) fused_arrive(&tm->cntA, &tm->genA, it, T);
        if constexpr (Cfg::PREFETCH) {
            if (p + NT < nplanes && member < (unsigned)UR) load_rows(p + NT, member, w);
        }
        if (tid == 0) shw[5] = fused_wait<LIMIT>(&tm->genA, it, ctl, FUSED_ERR_WAIT_A) ? 1u : 0u;
        __syncthreads();
        if (shw[5] == 0) return;
        // ---- columns of plane p ----
        bool arrived = false;
        for (unsigned u = member; u < (unsigned)UC; u += T) {
#pragma unroll
            for (int k = 0; k < EY; ++k) v[k] = fused_sload<Cfg::SLOAD, V>(Sr, (unsigned)((jy + TY * k) * PITCH + u * CB + cy));
            if (u + T >= (unsigned)UC) {  // my last tile: once it sits in registers the team may overwrite S
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) fused_arrive(&tm->cntB, &tm->genB, it, T);
                arrived = true;
            }
            if constexpr (Cfg::MATH) {
                if constexpr (Cfg::TW_RELOAD) {
                    V twl[TWNY > 0 ? TWNY : 1];
                    load_twiddles<V, PY, 0, DIR, TWPOW>(twl, twy, jy);
                    run_stages<V, PY, 0, DIR, CB, false, false, TW_REG, TWPOW>(v, twl, lds, jy, cy);
                } else {
                    run_stages<V, PY, 0, DIR, CB, false, false, TW_REG, TWPOW>(v, twyr, lds, jy, cy);
                }
            }
            V* op = out + (long long)(a_first + p) * otile.a_stride + (long long)u * CB * otile.b_stride;
#pragma unroll
            for (int k = 0; k < EY; ++k) gstore<Cfg::OUT_NT>(op + orel[k], v[k]);
        }
        if (!arrived && tid == 0) fused_arrive(&tm->cntB, &tm->genB, it, T);
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Split variant: the team's scratch holds only HALF a plane.  A wavefront Z-transforms the row PAIR (y, y + N1/2) and
// applies the first (radix-2, decimation-in-frequency) stage of the Y transform in registers:
//     a[y] = r[y] + r[y + N1/2]                     -> even outputs  Y[2k]   = DFT_{N1/2}(a)[k]
//     b[y] = (r[y] - r[y + N1/2]) * W_N1^{y}        -> odd  outputs  Y[2k+1] = DFT_{N1/2}(b)[k]
// The a half-plane goes through the scratch first (rows -> barrier A -> column tiles of N1/2 points -> even output rows)
// while b waits in registers, then b follows through the same scratch.  The L2 footprint of the exchange is halved
// (2 MiB per XCD for a 512 x 512 fp64 plane), which is what lets the exchanged lines survive next to the streamed input
// and output in a 4 MiB L2; the price is four team barriers per plane instead of two.
// PYH is the plan of the N1/2-point column transform.  Every workgroup owns at most one row unit (GR pairs) per plane:
// the host only uses this kernel when teams have at least N1 / (2 GR) members.
template <class V, class PZ, class PYH, int DIR, class Cfg>
__global__ void __attribute__((amdgpu_flat_work_group_size(Cfg::THREADS, Cfg::THREADS),
                               amdgpu_waves_per_eu(Cfg::MIN_WAVES > 0 ? Cfg::MIN_WAVES : 1)))
fused_yz_split_kernel(const V* in, V* out, V* scratch, FusedCtl* ctl, const V* __restrict__ twz, const V* __restrict__ twyh,
                      const V* __restrict__ twy, AxisMap omap, TileMap otile, long long in_plane_stride, unsigned nplanes,
                      unsigned a_first) {
    constexpr int THREADS = Cfg::THREADS;
    constexpr int N2 = PZ::N, NH = PYH::N, N1 = 2 * NH;
    constexpr int EZ = PZ::E, TZ = PZ::T, EY = PYH::E, TY = PYH::T;
    static_assert(THREADS % TZ == 0 && TZ <= 64 && 64 % TZ == 0, "rows: one FFT inside one wavefront");
    constexpr int GR = THREADS / TZ;  // row pairs per unit
    constexpr int CB = (int)(128 / sizeof(V));  // full cache lines per row segment
    constexpr int GT = CB * TY;                // threads per column tile
    static_assert(THREADS % GT == 0, "the workgroup is a whole number of column tiles");
    constexpr int GC = THREADS / GT;           // column tiles processed at once
    static_assert(NH % GR == 0 && N2 % (CB * GC) == 0, "plane must split into whole units");
    constexpr int UR = NH / GR, UC = N2 / (CB * GC);
    constexpr int PITCH = N2 + Cfg::SPAD;
    constexpr bool TWPOW = true;
    constexpr int  ROW_LDS = N2 + N2 / 8;
    constexpr unsigned LIMIT = Cfg::TIMEOUT_TICKS;

    extern __shared__ __attribute__((aligned(16))) char dfft_smem[];
    unsigned* shw = reinterpret_cast<unsigned*>(dfft_smem);
    V*        lds = reinterpret_cast<V*>(dfft_smem + 64);

    const int tid = threadIdx.x;
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 15u;
        const unsigned local = __hip_atomic_fetch_add(&ctl->xcc_count[xcc], 1u, __ATOMIC_RELAXED, DFFT_AGENT);
        const unsigned done = __hip_atomic_fetch_add(&ctl->registered, 1u, __ATOMIC_RELAXED, DFFT_AGENT) + 1u;
        bool ok = true;
        if (done != gridDim.x) ok = fused_wait<LIMIT>(&ctl->registered, gridDim.x, ctl, FUSED_ERR_REGISTER);
        unsigned nt = 0, rank = 0, T = 0;
        for (unsigned x = 0; x < 16; ++x) {
            const unsigned n = fused_ld(&ctl->xcc_count[x]);
            if (n == 0) continue;
            if (x == xcc) {
                rank = nt;
                T = n;
            }
            ++nt;
        }
        if (ok && T < (unsigned)UR) {  // a member would own two row units: not supported by this variant
            __hip_atomic_store(&ctl->error, (unsigned)FUSED_ERR_TEAM_SIZE, __ATOMIC_RELAXED, DFFT_AGENT);
            ok = false;
        }
        shw[0] = xcc;
        shw[1] = local;
        shw[2] = T;
        shw[3] = nt;
        shw[4] = rank;
        shw[5] = ok ? 1u : 0u;
    }
    __syncthreads();
    if (shw[5] == 0) return;
    const unsigned team = shw[0], member = shw[1], T = shw[2], NT = shw[3], rank = shw[4];
    FusedTeam*     tm = &ctl->team[team];
    V*             S = scratch + (size_t)team * NH * PITCH;
    FusedScratch<V> Sr;
    Sr.base = S;
    Sr.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)S, 0, (int)((size_t)NH * PITCH * sizeof(V)), 0x00020000);

    const int gz = tid / TZ, jz = tid % TZ;
    const int gc = tid / GT, tc = tid % GT, cy = tc % CB, jy = tc / CB;
    V*        lds_row = lds + gz * ROW_LDS;
    V*        lds_col = lds + (size_t)gc * NH * CB;
    const bool has_rows = member < (unsigned)UR;
    const int  yrow = (int)member * GR + gz;  // this group's row pair (yrow, yrow + NH)

    constexpr int TWNZ = TwTotal<PZ, TWPOW>::value, TWNY = TwTotal<PYH, TWPOW>::value;
    constexpr bool KEEP_TW = Cfg::MATH && !Cfg::TW_RELOAD;
    V twzr[KEEP_TW && TWNZ > 0 ? TWNZ : 1], twyr[KEEP_TW && TWNY > 0 ? TWNY : 1];
    if constexpr (KEEP_TW) {
        load_twiddles<V, PZ, 0, DIR, TWPOW>(twzr, twz, jz);
        load_twiddles<V, PYH, 0, DIR, TWPOW>(twyr, twyh, jy);
    }
    V wy = V{1, 0};  // W_N1^{yrow}
    if constexpr (Cfg::MATH) {
        if (has_rows) {
            wy = twy[yrow];
            if (DIR < 0) wy.y = -wy.y;
        }
    }
    // output offsets of the points this thread holds after a column transform: natural index k' = jy + TY*k of the
    // half-length transform lands in output row 2k' + h (skeleton builds: row k' + h*NH, the identity)
    unsigned orel[EY];
#pragma unroll
    for (int k = 0; k < EY; ++k) {
        const int idx = Cfg::MATH ? 2 * (jy + TY * k) : jy + TY * k;
        const int ob = idx / omap.blk;
        orel[k] = (unsigned)(block_term(omap, ob) + (idx - ob * omap.blk) * omap.stride + cy * omap.cstride);
    }
    // the odd rows / second half: + one row, or (two-level / multi-block maps) recomputed
    unsigned odelta[EY];
#pragma unroll
    for (int k = 0; k < EY; ++k) {
        const int idx = Cfg::MATH ? 2 * (jy + TY * k) + 1 : jy + TY * k + NH;
        const int ob = idx / omap.blk;
        odelta[k] = (unsigned)(block_term(omap, ob) + (idx - ob * omap.blk) * omap.stride + cy * omap.cstride) - orel[k];
    }

    V r1[EZ], r2[EZ];
    auto load_pair = [&](unsigned p, V* d1, V* d2) {
        const V* ip = in + (long long)(a_first + p) * in_plane_stride + (long long)yrow * N2 + jz;
#pragma unroll
        for (int k = 0; k < EZ; ++k) d1[k] = gload<Cfg::IN_NT>(ip + TZ * k);
#pragma unroll
        for (int k = 0; k < EZ; ++k) d2[k] = gload<Cfg::IN_NT>(ip + (long long)NH * N2 + TZ * k);
    };
    V w1[Cfg::PREFETCH ? EZ : 1], w2[Cfg::PREFETCH ? EZ : 1];
    if constexpr (Cfg::PREFETCH) {
        if (rank < nplanes && has_rows) load_pair(rank, w1, w2);
    }
    unsigned it = 0;  // barrier instance counter: two per plane (one per half)
    for (unsigned p = rank; p < nplanes; p += NT) {
        __syncthreads();
        if (has_rows) {
            if constexpr (Cfg::PREFETCH) {
#pragma unroll
                for (int k = 0; k < EZ; ++k) {
                    r1[k] = w1[k];
                    r2[k] = w2[k];
                }
            } else {
                load_pair(p, r1, r2);
            }
            if constexpr (Cfg::MATH) {
                auto zfft = [&](V* v) {
                    if constexpr (Cfg::TW_RELOAD) {
                        V twl[TWNZ > 0 ? TWNZ : 1];
                        load_twiddles<V, PZ, 0, DIR, TWPOW>(twl, twz, jz);
                        run_stages<V, PZ, 0, DIR, 1, true, true, TW_REG, TWPOW>(v, twl, lds_row, jz, 0);
                    } else {
                        run_stages<V, PZ, 0, DIR, 1, true, true, TW_REG, TWPOW>(v, twzr, lds_row, jz, 0);
                    }
                };
                zfft(r1);
                group_sync<true>();  // the wave's second row reuses the exchange buffer
                zfft(r2);
#pragma unroll
                for (int k = 0; k < EZ; ++k) {
                    const V a = cadd(r1[k], r2[k]);
                    const V d = csub(r1[k], r2[k]);
                    r1[k] = a;
                    r2[k] = cmul(d, wy);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            ++it;
            // S is free once every member has the previous half's columns in registers
            if (it > 1) {
                if (tid == 0) shw[5] = fused_wait<LIMIT>(&tm->genB, it - 1, ctl, FUSED_ERR_WAIT_B) ? 1u : 0u;
                __syncthreads();
                if (shw[5] == 0) return;
            }
            if (has_rows) {
                V* sp = S + (size_t)yrow * PITCH + jz;
#pragma unroll
                for (int k = 0; k < EZ; ++k) gstore<Cfg::SSTORE_NT>(sp + TZ * k, h == 0 ? r1[k] : r2[k]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) fused_arrive(&tm->cntA, &tm->genA, it, T);
            if constexpr (Cfg::PREFETCH) {
                if (h == 1 && p + NT < nplanes && has_rows) load_pair(p + NT, w1, w2);
            }
            if (tid == 0) shw[5] = fused_wait<LIMIT>(&tm->genA, it, ctl, FUSED_ERR_WAIT_A) ? 1u : 0u;
            __syncthreads();
            if (shw[5] == 0) return;
            bool arrived = false;
            for (unsigned u = member; u < (unsigned)UC; u += T) {
                const unsigned tile = u * GC + gc;
                V v[EY];
#pragma unroll
                for (int k = 0; k < EY; ++k) v[k] = fused_sload<Cfg::SLOAD, V>(Sr, (unsigned)((jy + TY * k) * PITCH + tile * CB + cy));
                if (u + T >= (unsigned)UC) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (tid == 0) fused_arrive(&tm->cntB, &tm->genB, it, T);
                    arrived = true;
                }
                if constexpr (Cfg::MATH) {
                    if constexpr (Cfg::TW_RELOAD) {
                        V twl[TWNY > 0 ? TWNY : 1];
                        load_twiddles<V, PYH, 0, DIR, TWPOW>(twl, twyh, jy);
                        run_stages<V, PYH, 0, DIR, CB, false, false, TW_REG, TWPOW>(v, twl, lds_col, jy, cy);
                    } else {
                        run_stages<V, PYH, 0, DIR, CB, false, false, TW_REG, TWPOW>(v, twyr, lds_col, jy, cy);
                    }
                }
                V* op = out + (long long)(a_first + p) * otile.a_stride + (long long)tile * CB * otile.b_stride;
#pragma unroll
                for (int k = 0; k < EY; ++k) gstore<Cfg::OUT_NT>(op + orel[k] + (h == 1 ? odelta[k] : 0u), v[k]);
            }
            if (!arrived && tid == 0) fused_arrive(&tm->cntB, &tm->genB, it, T);
        }
    }
}

template <class V, class PZ, class PYH, class Cfg> struct FusedSplitGeom {
    static constexpr int    GR = Cfg::THREADS / PZ::T;
    static constexpr int    CB = (int)(128 / sizeof(V));
    static constexpr int    GC = Cfg::THREADS / (CB * PYH::T);
    static constexpr size_t ROW_BYTES = (size_t)GR * (PZ::N + PZ::N / 8) * sizeof(V);
    static constexpr size_t COL_BYTES = (size_t)GC * PYH::N * CB * sizeof(V);
    static constexpr size_t LDS_BYTES = 64 + (ROW_BYTES > COL_BYTES ? ROW_BYTES : COL_BYTES);
    static constexpr size_t SCRATCH_ELEMS_PER_TEAM = (size_t)PYH::N * (PZ::N + Cfg::SPAD);
    static constexpr int    MIN_TEAM = PYH::N / GR;
};

template <class V, class PZ, class PY, class Cfg> struct FusedGeom {
    static constexpr int    GR = Cfg::THREADS / PZ::T;
    static constexpr int    CB = Cfg::THREADS / PY::T;
    static constexpr size_t ROW_BYTES = (size_t)GR * (PZ::N + PZ::N / 8) * sizeof(V);
    static constexpr size_t COL_BYTES = (size_t)PY::N * CB * sizeof(V);
    static constexpr size_t LDS_BYTES = 64 + (ROW_BYTES > COL_BYTES ? ROW_BYTES : COL_BYTES);
    static constexpr size_t SCRATCH_ELEMS_PER_TEAM = (size_t)PY::N * (PZ::N + Cfg::SPAD);
};

}  // namespace dfft
