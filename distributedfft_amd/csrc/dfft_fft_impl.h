// dfft_fft_impl.h -- the Stockham FFT kernel template for gfx950 (wave64, LDS exchange, register twiddles).
//
// One template serves every FFT on the slab pipeline's hot path:
//   t0 Z pass  : contiguous rows            (reference K0a, templateFFT.cpp:6085-6086, fft_mpi_3d_api.cpp:496-511)
//   t0 Y pass  : stride-N2 columns + fused t1 pack  (reference K0b + K1, kernel_func.cpp:73-86)
//   t3 X pass  : stride-(yl*N2) columns + fused transposing store (reference K3a + K3b, fft_mpi_3d_api.cpp:539-551)
// and their inverses.  It is NOT a translation of the generated reference kernels:
//   * each thread loads its E points straight from HBM in the pattern the first radix stage consumes
//     (idx = j + T*k), and the last stage leaves results in that same pattern, so a whole FFT needs only
//     (stages-1) LDS exchanges (reference: stages+1 LDS round trips, 2*stages barriers);
//   * a 512-point FFT is exactly one wave64 (64 lanes x 8 points): the row kernel needs no s_barrier at all;
//   * twiddles e^{-2 pi i r m / (Ns R)} are fetched once per persistent thread into VGPRs and reused for every
//     tile the block processes (reference: global LUT read per stage per FFT); only the powers 1, 2, 4 are kept, the
//     others are products (keeps the kernels under 128 VGPRs = 4 waves per SIMD);
//   * the column kernel keeps CB adjacent columns as the fastest LDS dimension, so every ds_write_b128 /
//     ds_read_b128 lane group is contiguous (bank-conflict free without padding) and every HBM access is a
//     full 128-byte line; pack (t1) and the X transpose are folded into the store address map.
//
// Stockham stage (Ns = product of earlier radices, butterfly id jq = j + q*T, q < E/R):
//   u[r]  = v[q + r*E/R] * w^{r*(jq mod Ns)},  w = e^{-2 pi i DIR/(Ns*R)}
//   u     = DFT_R(u)
//   dst   = (jq / Ns) * Ns * R + (jq mod Ns) + r*Ns        (scatter through LDS, skipped for the last stage)
#pragma once
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "dfft_butterfly.h"
#include "dfft_kernels.h"

namespace dfft {

template <int N_, int E_, int... Rs> struct Plan {
    static constexpr int N = N_, E = E_, T = N_ / E_, S = sizeof...(Rs);
    static constexpr int R[sizeof...(Rs)] = {Rs...};
    static_assert((Rs * ... * 1) == N_, "radix product must equal N");
    static_assert(((E_ % Rs == 0) && ...), "every radix must divide E");
    static_assert(N_ % E_ == 0, "E must divide N");
};

// Kernel tuning policy (compile time).  The defaults are what the library ships; the A/B harnesses of rounds 1-2 (removed; logs under
// profiles/r01/experiments, profiles/r02/experiments) instantiated others.
struct TuneDefault {
    static constexpr bool TWPOW = true;    // keep w^1, w^2, w^4 per butterfly, derive the other powers by products
    static constexpr bool OSTAGE = false;  // column kernel: stage results through LDS so every wave stores 1 KiB runs
    static constexpr bool NTL = false;     // non-temporal global loads
    static constexpr bool NTS = false;     // non-temporal global stores
    static constexpr int MIN_WAVES = 0;    // amdgpu_waves_per_eu lower bound (0 = let the compiler decide)
    static constexpr int CB_OVERRIDE = 0;  // columns per tile (0 = cols_per_tile())
    static constexpr bool PLAIN = false;   // single-block address maps: offsets = base + k*step (no per-point VGPRs)
    static constexpr bool PREFETCH = false; // issue the next tile's loads before processing the current one (+E complex VGPRs)
    static constexpr bool FULL_PREFETCH = false;  // PREFETCH also where the second register set costs 64 VGPRs (16 points of 16 bytes)
    static constexpr bool PLAIN_IN = false;  // single-block input map: offset = base + k * step (no per-point VGPRs on the load side)
    static constexpr bool EARLY_WAIT = false;  // wait for the prefetched tile BEFORE this tile's stores are issued (pin_loaded)
    static constexpr bool ROT = false;      // row rotation of an exchange-buffer side compiled in (RotMap, dfft_kernels.h)
    static constexpr int ROT_IN = 0, ROT_OUT = 0;  // RotMap mode of the input / output side (compile time: 0, 1 or 2)
};
// any policy + the row rotation of the exchange buffers on one side (RIN / ROUT = RotMap::in_mode / out_mode).  The mode is a
// compile-time property on purpose: a first version tested rm.in_mode / rm.out_mode at run time inside the unrolled load / store
// loops, and that build stored the results of points 5 and 7 of the 8-point-per-thread Y pass wrongly (GPU parity test
// test_rotated_exchange_rows_vs_oracle, 64^3 on 2 devices); with the mode in the type all 24 cases are bit-identical.
template <class Base, int RIN, int ROUT> struct WithRot : Base {
    static constexpr bool ROT = true;
    static constexpr int  ROT_IN = RIN, ROT_OUT = ROUT;
};

// Column kernel whose store side is the transposed one ([..][z][kx], kx fastest: the forward X pass).  Measured on
// MI355X at 512^3 fp64 (profiles/r01/experiments): direct 128-byte-segment stores 1.20 ms; results staged through LDS so
// each wave stores 1 KiB runs + two resident blocks per CU (<= 128 VGPRs) + streaming (non-temporal) access 0.92 ms.
// Adding a register prefetch of the next tile (the block then overlaps HBM latency with its own barriers) beats holding
// the kernel to 128 VGPRs for a second resident block: 0.95 -> 0.90 ms.
struct TuneTransposedStore {
    static constexpr bool TWPOW = true;
    static constexpr bool OSTAGE = true;
    static constexpr bool NTL = true;
    static constexpr bool NTS = true;
    static constexpr int MIN_WAVES = 0;
    static constexpr int CB_OVERRIDE = 0;
    static constexpr bool PLAIN = false;
    static constexpr bool PREFETCH = true;
    static constexpr bool FULL_PREFETCH = false;
    static constexpr bool PLAIN_IN = false;
    static constexpr bool EARLY_WAIT = false;
    static constexpr bool ROT = false;
    static constexpr int ROT_IN = 0, ROT_OUT = 0;
};
// The same kernel for 16 points per thread (1024-point X pass) with a WHOLE second register set.  What made room: the input side
// of the X pass is a single-block map (plane stride x FFT index), so the 16 per-point load offsets become one register plus
// k * step with a wave-uniform step (PLAIN_IN; kept out of the tile loop's invariants, see load_part), and the staged store's
// offsets were affine already -- 228 VGPRs like the half prefetch, 246 with rotated rows, no scratch
// (profiles/r03/kernel_resources.txt).  Measured: fp32 pairs 4.95 -> 5.47 TB/s, the rotated P = 8 form 4.41 -> 4.67, fp64 at
// P = 1 4.62 -> 4.68 (profiles/r03/experiments/variant_ab.log).  Why it matters: gfx9 counts loads and stores in ONE vmcnt, and with
// both kinds outstanding the compiler has to wait for all of them, so a tile's exposed memory phases are what is NOT in flight
// underneath its exchanges -- the half prefetch left two such phases per tile (the second half of the loads, then the stores)
// and gained nothing in fp64 (profiles/r03/experiments/half_prefetch_ab.log); with the whole next tile in flight underneath the
// exchanges only the store drain is left, the structure of the 512-point X pass.
struct TuneTransposedStoreFull : TuneTransposedStore {
    static constexpr bool FULL_PREFETCH = true;
    static constexpr bool PLAIN_IN = true;
};
// EARLY_WAIT: the same vmcnt rule seen from the other side.  A prefetching kernel ends its iteration with "v = vnext", and that
// copy has to wait for the prefetched loads -- with this tile's stores just issued the wait becomes vmcnt(0), i.e. the block
// drains its own stores before it starts the next tile's arithmetic.  Waiting for the prefetched registers BEFORE the stores are
// issued (they have had the whole tile's exchanges to arrive) lets the stores drain underneath the next tile instead.
struct TuneTransposedStoreFullEarly : TuneTransposedStoreFull {
    static constexpr bool EARLY_WAIT = true;
};
struct TuneTransposedStoreEarly : TuneTransposedStore {
    static constexpr bool EARLY_WAIT = true;
};

// Column kernel default: the next tile's loads are issued before the current tile's exchanges (0.945 -> 0.869 ms on the
// 512^3 fp64 Y pass; no gain for the barrier-free row kernel, which keeps TuneDefault).
struct TuneCols : TuneDefault {
    static constexpr bool PREFETCH = true;
};

// Column kernels whose two sides are single-block maps (single-GPU plans, the 1-D column API, the backward X pass): every offset
// is base + k * step with wave-uniform steps (PLAIN), and -- as in TuneTransposedStoreFull -- the per-point values are kept out of
// the tile loop's invariants, where the compiler would otherwise park 2 E of them in registers.  Instantiated for the lengths whose
// column kernels spilled with per-point offsets: 1280 (20 points per thread) 68 -> 0 B at 202 VGPRs, 1536 (24) 92 -> 0 B at 198,
// 1000 (800-thread blocks, 128-VGPR budget) 164 -> 24 B (profiles/r03/kernel_resources.txt).
template <class Base> struct Plain : Base {
    static constexpr bool PLAIN = true;
};

// A side of the launch whose COLUMNS are far apart in memory and whose FFT index is the contiguous one, accessed directly (no staged
// image): the inverse X pass of fp64 plans reads [..][z][kx] that way, ragged / un-stageable forward X passes store it.  There the
// lanes of 8 consecutive butterfly ids of one column form the 128-byte pieces, so the wave-interleaved labelling (ids 8 apart in a
// wavefront) would turn every piece into eight 16-byte accesses (measured: inverse X pass of 1024 x 768 x 512 fp64 2.46 -> 3.45 ms);
// these launches keep the tid / CB numbering.
template <class Base> struct TransposedSide : Base {
    static constexpr bool NO_OWNED = true;
};
template <class T, class = void> struct tune_no_owned : std::false_type {};
template <class T> struct tune_no_owned<T, std::void_t<decltype(T::NO_OWNED)>> : std::bool_constant<T::NO_OWNED> {};

// Cache-policy variants for the Infinity-Cache-blocked Z+Y stage (execute_forward/backward in dfft_plan.cpp).
struct TuneStreamIn : TuneDefault {
    static constexpr bool NTL = true;
};
struct TuneColsStreamIn : TuneCols {
    static constexpr bool NTL = true;
};
struct TuneColsStreamOut : TuneCols {
    static constexpr bool NTS = true;
};

// Per data type V: W = scalar complex of the twiddles, G = what one lane moves to/from HBM (one V), LANES = columns per V.
template <class V> struct VecTraits {
    using W = V;
    using G = V;
    static constexpr int LANES = 1;
    static __device__ __forceinline__ V from_g(G g) { return g; }
    static __device__ __forceinline__ G to_g(V v) { return v; }
    static __device__ __forceinline__ V zero() { return V{0, 0}; }
    static __device__ __forceinline__ W lane(V v, int) { return v; }
};
template <> struct VecTraits<cpair> {
    using W = float2;
    using G = f32x4;  // (re0, im0, re1, im1): two adjacent columns as they lie in memory
    static constexpr int LANES = 2;
    static __device__ __forceinline__ cpair from_g(G g) { return cpair{f32x2{g.x, g.z}, f32x2{g.y, g.w}}; }
    static __device__ __forceinline__ G to_g(cpair v) { return G{v.x.x, v.y.x, v.x.y, v.y.y}; }
    static __device__ __forceinline__ cpair zero() { return cpair{f32x2{0.f, 0.f}, f32x2{0.f, 0.f}}; }
    static __device__ __forceinline__ W lane(cpair v, int l) { return l == 0 ? W{v.x.x, v.y.x} : W{v.x.y, v.y.y}; }
};

// number of stored twiddle powers per butterfly
constexpr int tw_slots(int R, bool pow2only) { return !pow2only ? R - 1 : (R >= 5 ? 3 : (R >= 3 ? 2 : (R >= 2 ? 1 : 0))); }

// When T is a multiple of Ns all B butterflies of a thread's stage share one twiddle set ((j + q*T) mod Ns does not depend on
// q), so only SLOTS registers are needed for that stage.  Counting this way would move N = 100, 1024 and 2048 from the LDS / L2
// table to registers; measured (round 1, profiles/r01/experiments) that costs 3-6 % on the 1024/2048-point column kernels (register
// pressure), so the sharing is switched on only for 4096 points, where 16 points x 256 threads then keep 12 twiddles in
// registers instead of a 64 KiB LDS table that, next to the 72 KiB row tile, would leave one workgroup per CU.
#ifndef DFFT_TW_EFFECTIVE
#define DFFT_TW_EFFECTIVE 0
#endif
template <class P> constexpr bool tw_sharing() { return DFFT_TW_EFFECTIVE || P::N >= 4096; }

// Stage-major LDS twiddle table (the default since round 4; -DDFFT_TW_STAGE_MAJOR=0 builds the natural-order table for A/B runs): the
// LDS copy of the twiddle table (TW_LDS kernels: 16 / 24 points per thread -- 768, 1024, 2048 points) is laid out [stage][m][r - 1]
// like the reference's LUT (templateFFT.cpp:5120-5141) instead of as the natural N-entry table read at r * m * N / (Ns R).  In the
// early stages (small Ns) the lanes of a ds_read_b128 group read a handful of distinct entries a power-of-two number of bytes apart
// -- the same banks, 2-4-way conflicts on R - 1 reads per butterfly (SQ_LDS_BANK_CONFLICT 9-36 % of these kernels' LDS cycles,
// profiles/r03/experiments/lds_bank_conflicts_long_x_pass.log); stage-major they are consecutive, and the per-point index
// arithmetic shrinks (1024-point X pass 228 -> 220 VGPRs, paired 2048-point tiles 234 -> 212).  Same table values: bit-identical
// results (sha256 of eight 3D results, profiles/r04/experiments/lib_ab_stage_major_twiddles.log).  Measured, two processes each,
// interleaved: X pass of 2048 x 2048 x 1024 fp32 per rank at P = 8 (config 5) 1.947 -> 1.813 ms (4.41 -> 4.74 TB/s), 1024^3 fp32
// 3.17 -> 3.06, 2048 x 1024 x 512 fp64 7.11 -> 6.96, fp32 4.61 -> 4.54, 1024 x 768 x 512 fp64 2.729 -> 2.710, fp32 1.116 -> 1.098; t0 with
// a 768- / 1024-point Y axis -1 ... -2.5 %; nothing slower.  Entries: sum over stages s >= 1 of Ns (R - 1) = N - R_0 <= N: the
// same LDS area.
#ifndef DFFT_TW_STAGE_MAJOR
#define DFFT_TW_STAGE_MAJOR 1
#endif
template <class P, int S, bool TWPOW> struct StageInfo {
    using Prev = StageInfo<P, S - 1, TWPOW>;
    static constexpr int R = P::R[S];
    static constexpr int NS = Prev::NS * Prev::R;
    static constexpr int SM_OFF = Prev::SM_OFF + (S > 1 ? Prev::NS * (Prev::R - 1) : 0);  // stage-major LDS table: first entry of stage S
    static constexpr int B = P::E / R;
    static constexpr int SLOTS = tw_slots(R, TWPOW);
    static constexpr bool SHARED = tw_sharing<P>() && (P::T % NS == 0);  // one set for all B butterflies
    static constexpr int QSTRIDE = SHARED ? 0 : SLOTS;                   // register distance between butterflies' sets
    static constexpr int TWOFF = Prev::TWOFF + Prev::TWCNT;
    static constexpr int TWCNT = SHARED ? SLOTS : B * SLOTS;
};
template <class P, bool TWPOW> struct StageInfo<P, 0, TWPOW> {
    static constexpr int R = P::R[0];
    static constexpr int NS = 1;
    static constexpr int SM_OFF = 0;
    static constexpr int B = P::E / R;
    static constexpr int SLOTS = 0;
    static constexpr bool SHARED = false;
    static constexpr int QSTRIDE = 0;
    static constexpr int TWOFF = 0;
    static constexpr int TWCNT = 0;
};
template <class P, bool TWPOW> struct TwTotal {
    using L = StageInfo<P, P::S - 1, TWPOW>;
    static constexpr int value = L::TWOFF + L::TWCNT;
};

template <int CB, bool PAD> __device__ __forceinline__ int lds_index(int idx, int c) {
    // CB == 1 (row kernel): one pad element every 8 keeps the stride-R scatter of the first stage on
    // distinct banks for ds_write_b128 (stride 9*16 B) and ds_write_b64 (stride 9*8 B).
    const int p = PAD ? idx + (idx >> 3) : idx;
    return p * CB + c;
}

template <bool WAVE_LOCAL> __device__ __forceinline__ void group_sync() {
    if (WAVE_LOCAL) {
        // The whole FFT lives in one wavefront: DS operations of a wave are processed in issue order, so only
        // the compiler has to be kept from reordering across the exchange.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// stored slot s of a butterfly holds w^{1 << s} when TWPOW, else w^{s+1}
template <class W, class P, int S, int DIR, bool TWPOW>
__device__ __forceinline__ void load_twiddles(W* twr, const W* __restrict__ tw, int j) {
    if constexpr (S < P::S) {
        using SI = StageInfo<P, S, TWPOW>;
        if constexpr (S > 0) {
#pragma unroll
            for (int q = 0; q < (SI::SHARED ? 1 : SI::B); ++q) {
                const int m = (j + q * P::T) % SI::NS;
#pragma unroll
                for (int s = 0; s < SI::SLOTS; ++s) {
                    const int r = TWPOW ? (1 << s) : (s + 1);
                    W w = tw[(r * m) * (P::N / (SI::NS * SI::R))];
                    if (DIR < 0) w.y = -w.y;
                    twr[SI::TWOFF + q * SI::SLOTS + s] = w;
                }
            }
        }
        load_twiddles<W, P, S + 1, DIR, TWPOW>(twr, tw, j);
    }
}

// Wave-interleaved labelling of the butterfly ids of a column tile (round 6; see the wave-owned exchange in run_stages).  A tile of CB
// columns x T butterfly threads spans NW = CB T / 64 wavefronts.  Numbering the threads j = tid / CB puts eight CONSECUTIVE ids into a
// wavefront; here wavefront w takes the ids congruent to w modulo NW instead: j = w + NW * (lane / CB).  The 8 (16 in fp32) lanes of a
// column group still cover one full 128-byte line / all 32 LDS banks per access, and which rows of the tile a wavefront touches never
// mattered to HBM (every row is a line of its own), so nothing changes on the memory side -- but every exchange after the first then
// stays inside a wavefront.  -DDFFT_WAVE_OWNED=0 builds the tid / CB numbering (A/B).
#ifndef DFFT_WAVE_OWNED
#define DFFT_WAVE_OWNED 1
#endif
// Where it is used: tiles whose column groups are whole 128-byte lines of 16-byte elements (fp64, fp32 column pairs), one-phase
// exchanges, and not the staged transposing store of column pairs.  Measured before that rule (profiles/r06/experiments/
// lib_ab_wave_owned.log): the labelling puts the lanes of a ds_read_b128 group (lanes {0-3, 12-15, 20-27}, ...: four different column
// groups, MI355X_MICROARCH.md LDS table) on positions 8 apart instead of consecutive ones -- the same half of the 256-byte bank row,
// 2-way conflicts on every exchange read (cured by lds_pos below) -- and the 8-byte image writes of the pairs' staged store 64 bytes
// apart instead of 8 (4-way instead of 2-way: 1024-point X pass on pairs 2.82 -> 3.21 ms); 4-column tiles (the paired half-line tiles of
// the 2048-point X pass) have two column groups per 128 bytes and lose either way (1.87 -> 2.21 ms).
template <class V, int CB, int T, int PH = 1, bool PAIR_IMAGE = false> constexpr int owned_waves() {
    constexpr int GT = CB * T;
    return (DFFT_WAVE_OWNED && sizeof(V) == 16 && (CB * sizeof(V)) % 128 == 0 && !PAIR_IMAGE && PH == 1 && GT % 64 == 0 && GT / 64 > 1 && 64 % CB == 0) ? GT / 64 : 0;
}
// Position p of the exchange tile under the labelling: the two 128-byte halves of every 256-byte bank row are swapped in every second
// group of 8 positions, so that the column groups of a ds_read_b128 lane group -- positions 8 apart -- alternate between the halves.
// Writes move whole 128-byte slots (an 8-lane ds_write_b128 group stays contiguous).  Identity without the labelling.
template <int NW> __device__ __forceinline__ constexpr int lds_pos(int p) {
    if constexpr (NW > 1) return p ^ ((p >> 3) & 1);
    else return p;
}
// butterfly id of thread `tid` of a tile's thread group (groups start at multiples of 64 threads)
template <int CB, int NW> __device__ __forceinline__ int tile_j(int tid) {
    if constexpr (NW > 1) return (tid >> 6) + NW * ((tid & 63) / CB);
    else return tid / CB;
}

// Row m of a stage's block of the stage-major LDS twiddle table.  Under the wave-interleaved labelling the lanes of a wavefront hold
// ids 8 (NW) apart, so for a stage with NS > NW their rows m = jq mod NS would lie NW (R - 1) entries apart -- the same banks for every
// lane group.  The rows are therefore stored in the order the lanes ask for them: m -> (m mod NW) (NS / NW) + m / NW (consecutive lanes,
// consecutive rows, (R - 1) 16-byte entries = an odd number of bank groups apart).  Identity for NS <= NW and without the labelling.
template <int NW, int NS> __device__ __forceinline__ constexpr int tw_row(int m) {
    if constexpr (NW > 1 && NS > NW && NS % NW == 0) return (m % NW) * (NS / NW) + m / NW;
    else return m;
}
// LDS copy of the twiddle table for TW_LDS kernels in stage-major order
// (TWS: stride of the source table -- 2 when the stages of an N/2-point plan read the table of the N-point transform, fft_dif2_tiles_kernel)
template <class W, class P, int S, int DIR, int NW = 0, int TWS = 1>
__device__ __forceinline__ void fill_stage_major(W* dst, const W* __restrict__ tw, int tid, int threads) {
    if constexpr (S < P::S) {
        using SI = StageInfo<P, S, false>;
        if constexpr (S > 0) {
            constexpr int R = SI::R, NS = SI::NS, CNT = NS * (R - 1), STRIDE = TWS * P::N / (NS * R);
            for (int i = tid; i < CNT; i += threads) {
                const int m = i / (R - 1), r = i % (R - 1) + 1;
                W w = tw[r * m * STRIDE];
                if (DIR < 0) w.y = -w.y;
                dst[SI::SM_OFF + tw_row<NW, NS>(m) * (R - 1) + (r - 1)] = w;
            }
        }
        fill_stage_major<W, P, S + 1, DIR, NW, TWS>(dst, tw, tid, threads);
    }
}
// TWMODE: where a stage finds its twiddles.
//   TW_REG    per-thread set preloaded into VGPRs
//   TW_LDS    direction-adjusted N-entry table staged in LDS once per block
//   TW_GLOBAL N-entry table read through L1/L2 (only when the LDS is needed for the exchange tile, e.g. N = 2048)
enum { TW_REG = 0, TW_LDS = 1, TW_GLOBAL = 2 };

// Cross-lane form of a wave-owned exchange (round 6, experiment: -DDFFT_XLANE=1; VERDICT r05 Next-1c).  Under the wave-interleaved labelling
// with NW = 8 waves, 8-column tiles and T = 64 butterfly threads a wavefront holds the ids w + 8 g (g = lane / 8, column = lane % 8).
// For a stage with NS R = 64 = T (NS = 8 n) the scatter / gather sends the result (g, q, r) to lane group (r, g mod n) as point
// g / n + R q: for each of the thread's B butterflies a TRANSPOSE between the upper lane-group bits and the register index r of 16-byte
// elements, followed by a renaming of registers.  gfx950 does a round of such a transpose on a register pair without selects:
// v_permlane32_swap (upper half of a <-> lower half of b), v_permlane16_swap (odd rows of a <-> even rows of b); lane bit 3 takes two
// v_mov_b32_dpp row_ror:8 with bank masks.  Cases: the exchange between the radix-8 stages of 512 / 1024 / 2048 = 8 8 ... (three rounds),
// the one behind the third radix-4 stage of 768 = 4 4 4 4 3 (NS = 16: two rounds, no DPP).  Pure data movement: bit-identical results.
// tools/xlane_probe.hip measures the exchange alone (profiles/r06/README.md section 1i).
#ifndef DFFT_XLANE
#define DFFT_XLANE 0
#endif
typedef unsigned xl_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned xl_u32x4 __attribute__((ext_vector_type(4)));
template <int NW, int CB, int T, int NS, int R, int S, class V> constexpr bool xlane_exchange() {
    return DFFT_XLANE && NW == 8 && CB == 8 && T == 64 && S > 0 && sizeof(V) == 16 && NS % 8 == 0 && NS * R == 64 && (R == 8 || R == 4 || R == 2);
}
// one round on a register pair: lanes whose bit BIT is clear keep a and receive the partner's a into b, the others keep b and receive the
// partner's b into a
template <int BIT, class V> __device__ __forceinline__ void xlane_round(V& a, V& b) {
    xl_u32x4 x, y;
    __builtin_memcpy(&x, &a, 16);
    __builtin_memcpy(&y, &b, 16);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if constexpr (BIT == 5) {
            const xl_u32x2 r = __builtin_amdgcn_permlane32_swap(x[d], y[d], false, false);
            x[d] = r.x;
            y[d] = r.y;
        } else if constexpr (BIT == 4) {
            const xl_u32x2 r = __builtin_amdgcn_permlane16_swap(x[d], y[d], false, false);
            x[d] = r.x;
            y[d] = r.y;
        } else {
            static_assert(BIT == 3, "lane bits 3..5");
            const unsigned nx = __builtin_amdgcn_update_dpp(x[d], y[d], 0x128 /* row_ror:8 */, 0xf, 0xc, false);
            const unsigned ny = __builtin_amdgcn_update_dpp(y[d], x[d], 0x128, 0xf, 0x3, false);
            x[d] = nx;
            y[d] = ny;
        }
    }
    __builtin_memcpy(&a, &x, 16);
    __builtin_memcpy(&b, &y, 16);
}
template <int BIT0, int RB, int R, int B, class V> __device__ __forceinline__ void xlane_transpose(V* v, int q) {
    // register bit RB of r <-> lane bit BIT0 + RB
    if constexpr (RB >= 0) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (!(r & (1 << RB))) xlane_round<BIT0 + RB>(v[q + r * B], v[q + (r | (1 << RB)) * B]);
        xlane_transpose<BIT0, RB - 1, R, B>(v, q);
    }
}

// PH = 2: the tile (N x CB elements) is twice what the LDS holds, so every exchange runs in two phases -- first the threads
// of columns [0, CB/2), then those of [CB/2, CB) -- through one half-size buffer.  HBM accesses keep full 128-byte lines
// (all CB columns of a row segment are loaded/stored together); only the LDS issue slots double.
// TWS: stride of the twiddle table behind twr in TW_LDS / TW_GLOBAL mode (2: the table of a transform twice as long)
// NW: waves per tile under the wave-interleaved labelling of the butterfly ids (wave_owned_j below), 0 = threads numbered tid / CB
// LOCALX: thread-local exchanges are renamings (below); false keeps them in LDS for kernels that have no register to spare
template <class V, class P, int S, int DIR, int CB, bool PAD, bool WAVE_LOCAL, int TWMODE, bool TWPOW, int PH = 1, int TWS = 1, int NW = 0, bool LOCALX = true>
__device__ __forceinline__ void run_stages(V* v, const typename VecTraits<V>::W* twr, V* lds, int j, int c) {
    using SI = StageInfo<P, S, TWPOW>;
    using W = typename VecTraits<V>::W;
    constexpr int R = SI::R, B = SI::B, NS = SI::NS, T = P::T, E = P::E;
#pragma unroll
    for (int q = 0; q < B; ++q) {
        V u[R];
#pragma unroll
        for (int r = 0; r < R; ++r) u[r] = v[q + r * B];
        if constexpr (S > 0) {
#if DFFT_TW_STAGE_MAJOR
            if constexpr (TWMODE == TW_LDS && TWS == 1) {
                const W* ts = twr + SI::SM_OFF + tw_row<NW, NS>((j + q * T) % NS) * (R - 1);
#pragma unroll
                for (int r = 1; r < R; ++r) u[r] = cmul(u[r], ts[r - 1]);
            } else
#endif
            if constexpr (TWMODE == TW_LDS) {
                const int m = ((j + q * T) % NS) * (TWS * P::N / (NS * R));
#pragma unroll
                for (int r = 1; r < R; ++r) u[r] = cmul(u[r], twr[r * m]);
            } else if constexpr (TWMODE == TW_GLOBAL) {
                const int m = ((j + q * T) % NS) * (TWS * P::N / (NS * R));
#pragma unroll
                for (int r = 1; r < R; ++r) {
                    W w = twr[r * m];
                    if (DIR < 0) w.y = -w.y;
                    u[r] = cmul(u[r], w);
                }
            } else if constexpr (TWPOW) {
                const W* ws = twr + SI::TWOFF + q * SI::QSTRIDE;
                W w[R > 1 ? R : 2];
                w[1] = ws[0];
                if constexpr (R >= 3) w[2] = ws[1];
                if constexpr (R >= 4) w[3] = cmul(w[1], w[2]);
                if constexpr (R >= 5) w[4] = ws[2];
                if constexpr (R >= 6) w[5] = cmul(w[4], w[1]);
                if constexpr (R >= 7) w[6] = cmul(w[4], w[2]);
                if constexpr (R >= 8) w[7] = cmul(w[4], w[3]);
#pragma unroll
                for (int r = 1; r < R; ++r) u[r] = cmul(u[r], w[r]);
            } else {
#pragma unroll
                for (int r = 1; r < R; ++r) u[r] = cmul(u[r], twr[SI::TWOFF + q * SI::QSTRIDE + (r - 1)]);
            }
        }
        Butterfly<R, DIR, V>::run(u);
#pragma unroll
        for (int r = 0; r < R; ++r) v[q + r * B] = u[r];
    }
    if constexpr (S + 1 < P::S) {
#ifdef DFFT_DBG_NOEXCH
        // measurement builds only (-DDFFT_DBG_NOEXCH): skip the LDS exchange to see the HBM + VALU time alone (wrong results)
        run_stages<V, P, S + 1, DIR, CB, PAD, WAVE_LOCAL, TWMODE, TWPOW, PH, TWS, NW, LOCALX>(v, twr, lds, j, c);
        return;
#endif
        // Thread-local exchange (round 6).  The scatter after stage S sends the result (jq, r) to position
        //     p = (jq / NS) NS R + jq mod NS + r NS,         jq = j + q T,
        // and position p is read back by thread p mod T as its point p / T.  When T divides NS every term but j is a multiple of
        // T: p mod T = j -- the thread reads back exactly what it wrote, the "exchange" is a renaming of its own registers
        //     v'[(qT / NS)(NS / T) R + (qT mod NS) / T + r NS / T] = v[q + r B]
        // (compile-time indices: no instruction at all), and the LDS round trip with its two barriers disappears.  True for the
        // stages whose remaining radix product fits the E points of a thread: the exchange in front of the last stage of 1024 = 8 8 8 2
        // on 16 x 64 threads (config 4's X axis, the halves of config 5's DIF-split 2048-point Y axis), of 768 = 4 4 4 4 3 on 12 x 64
        // (config 4's Y axis), 1536 = 8 8 8 3, the one-wavefront rows of 2048 points (32 x 64).  Same values in the same order of
        // operations: bit-identical results.  -DDFFT_LOCAL_EXCHANGE=0 builds the LDS form (A/B).
#ifndef DFFT_LOCAL_EXCHANGE
#define DFFT_LOCAL_EXCHANGE 1
#endif
        if constexpr (DFFT_LOCAL_EXCHANGE && LOCALX && NS % T == 0) {
            V t[E];
#pragma unroll
            for (int q = 0; q < B; ++q)
#pragma unroll
                for (int r = 0; r < R; ++r) t[((q * T) / NS) * (NS / T) * R + ((q * T) % NS) / T + r * (NS / T)] = v[q + r * B];
#pragma unroll
            for (int k = 0; k < E; ++k) v[k] = t[k];
        } else if constexpr (PH == 1 && xlane_exchange<NW, CB, T, NS, R, S, V>()) {
            constexpr int LR = R == 8 ? 3 : (R == 4 ? 2 : 1);  // bits of r; the lane bits are the top LR bits of the lane group: 6 - LR .. 5
            V t[E];
#pragma unroll
            for (int q = 0; q < B; ++q) {
                xlane_transpose<6 - LR, LR - 1, R, B>(v, q);
#pragma unroll
                for (int i = 0; i < R; ++i) t[i + R * q] = v[q + i * B];
            }
#pragma unroll
            for (int k = 0; k < E; ++k) v[k] = t[k];
        } else if constexpr (PH == 1) {
            // Wave-owned exchange (round 6).  Under the wave-interleaved labelling (wave of butterfly id j = j mod NW, wave_owned_j) a
            // thread READS positions j + T k of the tile in every exchange -- positions congruent to its wave's index modulo NW (NW
            // divides T) -- and WRITES positions congruent to jq mod NS modulo NW whenever NW divides NS: its own wave's again.  From
            // the first exchange with NW | NS on, every wave therefore reads and writes only its own residue class of the tile: no other
            // wave touches it, DS operations of one wave execute in issue order, and the two workgroup barriers of the exchange shrink to
            // compiler fences.  With radix-8 first stages and 8 waves per tile that is every exchange but the first: 512 = 8 8 8 keeps 2
            // of 4 barriers per tile, 1024 = 8 8 [8 2] 2 of 4, 2048 = 8 8 8 4 2 of 6 (plus the two of a staged store).  The labelling
            // changes which lane computes a butterfly, not what is computed: bit-identical results.
            constexpr bool OWNED = NW > 1 && S > 0 && NS % NW == 0 && T % NW == 0;
            constexpr bool WL = WAVE_LOCAL || OWNED;
            if constexpr (S > 0 || !WAVE_LOCAL) group_sync<WL>();  // WAR: previous readers are done
            // Padded rows (one element per 8): where the step between a thread's accesses is a multiple of 8 elements the
            // padded index is affine in the access number -- written out that way so that the accesses become ONE address
            // register plus immediate offsets (left to itself the compiler keeps an address register per access alive across
            // the tile loop: 128 of them for 16 points x 4 stages, which made the 4096-point row kernel spill).
            constexpr bool WAFF = PAD && (NS % 8 == 0 || (NS == 1 && R == 8));
            constexpr bool RAFF = PAD && (T % 8 == 0);
#pragma unroll
            for (int q = 0; q < B; ++q) {
                const int jq = j + q * T;
                const int base = (jq / NS) * (NS * R) + (jq % NS);
                if constexpr (WAFF) {
                    const int pb = lds_index<CB, PAD>(base, c);
                    constexpr int step = (NS % 8 == 0 ? NS + NS / 8 : 1) * CB;
#pragma unroll
                    for (int r = 0; r < R; ++r) lds[pb + r * step] = v[q + r * B];
                } else if constexpr (NW > 1) {
                    // positions lds_pos(base + r NS) written as ONE or TWO base registers plus compile-time offsets (left as the XOR
                    // the compiler keeps an address register per access: the 16-point column kernels spill)
                    if constexpr (NS % 16 == 0) {  // bit 3 of the position comes from base alone
                        const int b = lds_pos<NW>(base) * CB + c;
#pragma unroll
                        for (int r = 0; r < R; ++r) lds[b + r * NS * CB] = v[q + r * B];
                    } else if constexpr ((NS == 8 && R % 2 == 0) || (NS == 4 && R == 4)) {  // ... from r alone: r odd (NS = 8), r >= 2 (NS = 4)
                        const int b0 = base * CB + c, b1 = (base ^ 1) * CB + c;
#pragma unroll
                        for (int r = 0; r < R; ++r) lds[((NS == 8 ? r & 1 : (r >> 1) & 1) ? b1 : b0) + r * NS * CB] = v[q + r * B];
                    } else if constexpr (NS == 1 && (R == 8 || R == 4)) {  // ... from base, bit 0 from r: r ^ bit = r + bit (r even), r - bit (r odd)
                        const int bit = (base >> 3) & 1, b0 = (base + bit) * CB + c, b1 = (base - bit) * CB + c;
#pragma unroll
                        for (int r = 0; r < R; ++r) lds[((r & 1) ? b1 : b0) + r * CB] = v[q + r * B];
                    } else {
#pragma unroll
                        for (int r = 0; r < R; ++r) lds[lds_pos<NW>(base + r * NS) * CB + c] = v[q + r * B];
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) lds[lds_index<CB, PAD>(base + r * NS, c)] = v[q + r * B];
                }
            }
            group_sync<WL>();
            if constexpr (RAFF) {
                const int pj = lds_index<CB, PAD>(j, c);
#pragma unroll
                for (int k = 0; k < E; ++k) v[k] = lds[pj + k * ((T + T / 8) * CB)];
            } else if constexpr (NW > 1 && T % 16 == 0) {
                const int pj = lds_pos<NW>(j) * CB + c;  // (T k does not reach bit 3)
#pragma unroll
                for (int k = 0; k < E; ++k) v[k] = lds[pj + k * T * CB];
            } else {
#pragma unroll
                for (int k = 0; k < E; ++k) v[k] = lds[lds_index<CB, PAD>(lds_pos<NW>(j + T * k), c)];
            }
        } else {
            static_assert(PH == 1 || (!PAD && !WAVE_LOCAL && CB % PH == 0), "two-phase exchange: block-wide column tiles");
            constexpr int CH = CB / PH;
            const int     mine = c / CH, cl = c - mine * CH;
#pragma unroll
            for (int ph = 0; ph < PH; ++ph) {
                __syncthreads();  // WAR: the readers of the previous phase / stage are done
                if (mine == ph) {
#pragma unroll
                    for (int q = 0; q < B; ++q) {
                        const int jq = j + q * T;
                        const int base = (jq / NS) * (NS * R) + (jq % NS);
#pragma unroll
                        for (int r = 0; r < R; ++r) lds[(base + r * NS) * CH + cl] = v[q + r * B];
                    }
                }
                __syncthreads();
                if (mine == ph) {
#pragma unroll
                    for (int k = 0; k < E; ++k) v[k] = lds[(j + T * k) * CH + cl];
                }
            }
        }
        run_stages<V, P, S + 1, DIR, CB, PAD, WAVE_LOCAL, TWMODE, TWPOW, PH, TWS, NW, LOCALX>(v, twr, lds, j, c);
    }
}

template <class V, class P, int CB, int G, class Tune> struct KernelGeom {
    static constexpr int GT = CB * P::T;  // threads cooperating on one tile
    static constexpr int THREADS = GT * G;
    static constexpr bool WAVE_LOCAL = (GT <= 64) && (64 % GT == 0);
    static constexpr bool PAD = (CB == 1) && (P::N >= 16);
    static constexpr bool OSTAGE = Tune::OSTAGE && CB > 1 && P::S > 1;
    // exchange tile; the staged store needs [CB][N + 1] (one pad element per column keeps the transposed writes on
    // distinct banks)
    // a tile above 128 KiB (1280-, 1536-, 2048-point full-line column tiles) goes through the LDS in two phases of CB/2
    // columns each (run_stages): full 128-byte lines on the HBM side without a 256 KiB exchange buffer
    static constexpr int PH = (P::S > 1 && CB > 1 && (size_t)P::N * CB * sizeof(V) > 128 * 1024) ? 2 : 1;
    static_assert(CB % PH == 0 && P::E % PH == 0, "two-phase tiles need an even column count and an even E");
    // wave-interleaved butterfly ids (0: tid / CB).  Not for the plain column kernels with per-point rotated offsets (the rotated side of a
    // P > 1 plan's X pass without a staged store): they sit at the register limit (see LOCALX) and the second base register of the
    // swizzled exchange tips the 768-point ones over
    static constexpr bool ROT2_PLAIN = Tune::ROT && (Tune::ROT_IN == 2 || Tune::ROT_OUT == 2) && !OSTAGE;
    static constexpr int  NW = (ROT2_PLAIN || tune_no_owned<Tune>::value) ? 0 : owned_waves<V, CB, P::T, PH, Tune::OSTAGE && VecTraits<V>::LANES == 2>();
    // 16 points of 16 bytes per thread with per-point rotated offsets and no staged store (the rotated side of a P > 1 plan's backward
    // X pass at 1024 points): these kernels sit AT 256 registers, and with the last two stages back to back (thread-local exchange) the
    // compiler goes 28-44 bytes over -- they keep that exchange in LDS (kernel_resources: 0 B either way in round 5's form)
    // (likewise the one-wavefront rows of 2048 fp64 points, 32 points = 128 registers of data per thread: 400 -> 440 registers and
    // 0.42 -> 0.455 ms per GiB with the exchange in registers, profiles/r06/experiments/long_axis_kernels_ab.txt)
    // (and 729 points -- 648-thread workgroups, 168 registers -- with per-point rotated offsets: 20-44 bytes over with three of the
    // six radix-3 stages back to back; see also LOCALX_GENERAL in the kernel)
#ifndef DFFT_729_LOCALX
#define DFFT_729_LOCALX 0
#endif
    static constexpr bool LOCALX = !(ROT2_PLAIN && P::E * (int)sizeof(V) / 4 >= 64) && !(CB == 1 && P::E * (int)sizeof(V) / 4 >= 128) &&
                                   (DFFT_729_LOCALX || !(ROT2_PLAIN && (P::N == 729 || (P::N == 768 && VecTraits<V>::LANES == 2 && Tune::ROT_IN == 2))));
    // (768-point column pairs loading rotated points without a staged store -- an un-fused or natural-order forward X pass: 12 bytes over)
    static constexpr int EX_ELEMS = (P::S > 1) ? (PAD ? P::N + P::N / 8 : P::N) * CB / PH : 0;
    // staged image: one scalar column per row of N + OPAD twiddle-typed elements (OPAD = 2 keeps cpair rows 16-B aligned)
    static constexpr int LANES = VecTraits<V>::LANES;
    static constexpr int OPAD = LANES == 2 ? 2 : 1;
    static constexpr int OS_ELEMS = OSTAGE ? (P::N + OPAD) * CB / PH : 0;  // in units of V (= LANES scalar elements)
    static_assert(!OSTAGE || P::N % LANES == 0, "the staged store of column pairs needs an even length");
    static constexpr int LDS_ELEMS = EX_ELEMS > OS_ELEMS ? EX_ELEMS : OS_ELEMS;
    static constexpr int TWN = TwTotal<P, Tune::TWPOW>::value;
    static constexpr int TWLIVE = TWN;  // distinct twiddle sets a thread holds
    // Twiddles live in VGPRs when the per-thread set is small (<= 16 distinct complex); otherwise in an LDS copy of the
    // table, unless that would push the block past the 160 KiB of a CU (then they are read through L1/L2).  (Budget raised
    // from 128 KiB in round 2: with the table in LDS the 1000-point column kernels spill 150 B instead of 550 B and run at
    // 3.1 instead of 2.1 TB/s; the 2048-point half-line tile + table use exactly 160 KiB.)
    static constexpr int TWMODE = TWLIVE <= 16 ? TW_REG
                                  : ((size_t)LDS_ELEMS * G * sizeof(V) + P::N * sizeof(typename VecTraits<V>::W) <= 160 * 1024
                                         ? TW_LDS : TW_GLOBAL);
    static constexpr int TW_ELEMS = TWMODE == TW_LDS ? P::N : 0;  // in units of W
    // rounded up so that the exchange tile behind the twiddle copy stays 16-byte aligned (odd lengths in fp32)
    static constexpr size_t TW_BYTES = ((size_t)TW_ELEMS * sizeof(typename VecTraits<V>::W) + 15) / 16 * 16;
    static constexpr size_t LDS_BYTES = (size_t)LDS_ELEMS * G * sizeof(V) + TW_BYTES;
};

template <class V> struct native_vec;
template <> struct native_vec<double2> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct native_vec<float2> { typedef float type __attribute__((ext_vector_type(2))); };

template <bool NT> __device__ __forceinline__ f32x4 gload(const f32x4* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT> __device__ __forceinline__ void gstore(f32x4* p, f32x4 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <bool NT, class V> __device__ __forceinline__ V gload(const V* p) {
    if constexpr (NT) {
        using NV = typename native_vec<V>::type;
        const NV t = __builtin_nontemporal_load(reinterpret_cast<const NV*>(p));
        return V{t.x, t.y};
    } else {
        return *p;
    }
}
template <bool NT, class V> __device__ __forceinline__ void gstore(V* p, V v) {
    if constexpr (NT) {
        using NV = typename native_vec<V>::type;
        NV t;
        t.x = v.x;
        t.y = v.y;
        __builtin_nontemporal_store(t, reinterpret_cast<NV*>(p));
    } else {
        *p = v;
    }
}

// "these registers are needed now": an empty asm that reads the value and clobbers memory -- the compiler has to complete the
// loads that fill it before this point, and may not move later stores above it.  (Input only: an in/out operand would also pin
// the value to an architectural VGPR from here on, and kernels at the 256-register limit keep part of the prefetched set in AGPRs.)
__device__ __forceinline__ void pin_loaded(const double2& x) { asm volatile("" : : "v"(x.x), "v"(x.y) : "memory"); }
__device__ __forceinline__ void pin_loaded(const float2& x) { asm volatile("" : : "v"(x.x), "v"(x.y) : "memory"); }
__device__ __forceinline__ void pin_loaded(const cpair& x) { asm volatile("" : : "v"(x.x), "v"(x.y) : "memory"); }

// register prefetch only where the second register set is cheap (64 VGPRs, i.e. E = 16 fp64: 1024-point columns
// 4.6 -> 3.7 TB/s, round 1)
#ifndef DFFT_PREFETCH_MAX_REGS
#define DFFT_PREFETCH_MAX_REGS 32
#endif
// offset of block ib of an axis map (single- or two-level, see AxisMap)
__device__ __forceinline__ long long block_term(const AxisMap& m, int ib) {
    if (m.sub > 1) return (long long)(ib / m.sub) * m.blk_stride + (long long)(ib % m.sub) * m.sub_stride;
    return (long long)ib * m.blk_stride;
}
// host twin of block_term + the largest element offset an axis map produces for n points (kernels that keep their
// wave-uniform offsets in 32 bits are only chosen when this fits)
inline long long axis_max_offset(const AxisMap& m, int n) {
    long long mx = 0;
    const int nb = m.blk > 0 ? (n + m.blk - 1) / m.blk : 1;
    for (int ib = 0; ib < nb; ++ib) {
        const long long bt = m.sub > 1 ? (long long)(ib / m.sub) * m.blk_stride + (long long)(ib % m.sub) * m.sub_stride : (long long)ib * m.blk_stride;
        const int       cnt = (ib + 1) * m.blk <= n ? m.blk : n - ib * m.blk;
        const long long o = bt + (long long)(cnt - 1) * m.stride;
        if (o > mx) mx = o;
    }
    return mx;
}
// GENERAL = ragged last column tile and/or uneven last slab (slow-path address terms compiled in).
// All offsets, strides and column counts are in units of one V (for cpair: 16 bytes = two fp32 columns).
template <class V, class P, int CB, int G, int DIR, bool GENERAL, class Tune>
__global__ void __attribute__((amdgpu_flat_work_group_size(1, KernelGeom<V, P, CB, G, Tune>::THREADS),
                               amdgpu_waves_per_eu(Tune::MIN_WAVES > 0 ? Tune::MIN_WAVES : 1)))
fft_tiles_kernel(const typename VecTraits<V>::G* in, typename VecTraits<V>::G* out,
                 const typename VecTraits<V>::W* __restrict__ tw, AxisMap imap, AxisMap omap, TileMap itile, TileMap otile,
                 unsigned ntiles, unsigned tiles_per_a, int ncols, unsigned a_first, double scale, RotMap rm) {
    using KG = KernelGeom<V, P, CB, G, Tune>;
    using VT = VecTraits<V>;
    using W = typename VT::W;
    using GV = typename VT::G;
    constexpr int E = P::E, T = P::T, GT = KG::GT, N = P::N, LANES = VT::LANES;
    extern __shared__ __attribute__((aligned(16))) char dfft_smem[];
    W* ldstw = reinterpret_cast<W*>(dfft_smem);
    V* lds = reinterpret_cast<V*>(dfft_smem + KG::TW_BYTES) + (threadIdx.x / GT) * KG::LDS_ELEMS;

    const int g = threadIdx.x / GT;
    const int tid = threadIdx.x - g * GT;
    const int c = tid % CB;
    const int j = tile_j<CB, KG::NW>(tid);

    constexpr int TWN = KG::TWMODE == TW_REG ? KG::TWN : 0;
    W twreg[TWN > 0 ? TWN : 1];
    const W* twr = twreg;
    if constexpr (KG::TWMODE == TW_GLOBAL) {
        twr = tw;
    } else if constexpr (KG::TWMODE == TW_LDS) {
#if DFFT_TW_STAGE_MAJOR
        fill_stage_major<W, P, 0, DIR, KG::NW>(ldstw, tw, (int)threadIdx.x, KG::THREADS);
#else
        for (int i = threadIdx.x; i < N; i += KG::THREADS) {
            W w = tw[i];
            if (DIR < 0) w.y = -w.y;
            ldstw[i] = w;
        }
#endif
        __syncthreads();
        twr = ldstw;
    } else {
        load_twiddles<W, P, 0, DIR, Tune::TWPOW>(twreg, tw, j);
    }
    constexpr bool TWPOW = Tune::TWPOW && KG::TWMODE == TW_REG;

    // Per-thread element offsets of its E points relative to the tile base (constant over tiles).
    // PLAIN maps (one block, no uneven slab): offset = base + k * step with a wave-uniform step, no per-point VGPRs.
    constexpr bool PLAIN = Tune::PLAIN && !GENERAL && !KG::OSTAGE && !Tune::ROT;
    constexpr bool PLAIN_IN = PLAIN || (Tune::PLAIN_IN && !GENERAL);
    constexpr int NREL = PLAIN ? 1 : E;
    constexpr int ON = N / LANES;  // staged store: memory elements (GV) per scalar column
    // staged store with ON a multiple of the group size (every power-of-two plan): the element a thread stores at step k is
    // tid + const_k in scalar column const'_k, so its offset is tid + C1_k + C2_k * cstride -- no per-point register
    // (and likewise when the group size is a multiple of ON: column tid / ON + const_k, element tid % ON)
    constexpr bool OAFFINE = KG::OSTAGE && (ON % GT == 0 || GT % ON == 0);
    unsigned irel[PLAIN_IN ? 1 : E], orel[OAFFINE ? 1 : NREL];
    unsigned ilast = 0, olast = 0;
    const unsigned istep = (unsigned)(T * imap.stride), ostep = (unsigned)(T * omap.stride);
    if constexpr (PLAIN_IN) irel[0] = (unsigned)(j * imap.stride + c * imap.cstride);
    if constexpr (PLAIN) orel[0] = (unsigned)(j * omap.stride + c * omap.cstride);
#pragma unroll
    for (int k = 0; k < (PLAIN ? 0 : E); ++k) {
        const int idx = j + T * k;
        if constexpr (!PLAIN_IN) {
            const int ib = idx / imap.blk;
            irel[k] = (unsigned)(block_term(imap, ib) + (idx - ib * imap.blk) * imap.stride + c * imap.cstride);
            if (GENERAL && ib == imap.nblk - 1) ilast |= 1u << k;
        }
        // staged store: thread owns the linear memory elements tid + GT*k of the [CB*LANES scalar columns][ON] result
        // tile (omap.cstride is then the distance between SCALAR columns, omap.stride the one between memory elements)
        // (two-phase tiles: points [0, E/2) belong to the image of columns [0, CB/2), the rest to the second image)
        if constexpr (!OAFFINE) {
            constexpr int EPH = E / KG::PH;
            const int lin_o = tid + GT * (k % EPH);
            const int oc = KG::OSTAGE ? (k / EPH) * (CB / KG::PH) * LANES + lin_o / ON : c;
            const int oidx = KG::OSTAGE ? lin_o % ON : idx;
            const int ob = oidx / omap.blk;
            orel[k] = (unsigned)(block_term(omap, ob) + (oidx - ob * omap.blk) * omap.stride + oc * omap.cstride);
            if (GENERAL && ob == omap.nblk - 1) olast |= 1u << k;
        }
    }
    // offset of staged-store step k (k is a compile-time constant after unrolling)
    const long long obase = OAFFINE ? (ON % GT == 0 ? (long long)tid : (long long)(tid % ON) + (long long)(tid / ON) * omap.cstride) : 0;
    auto ostage_off = [&](int k) -> long long {
        if constexpr (OAFFINE) {
            constexpr int EPH = E / KG::PH;
            const int kl = k % EPH;
            const int oc = (k / EPH) * (CB / KG::PH) * LANES + (GT * kl) / ON;  // exact in both cases
            return obase + (ON % GT == 0 ? (GT * kl) % ON : 0) + (long long)oc * omap.cstride;
        } else {
            return (long long)orel[k];
        }
    };

    const unsigned tstep = gridDim.x * G;
    // XCD-aware tile order for tiles narrower than a cache line (long FFTs whose full-line tile does not fit the LDS): the Q
    // tiles that share the 128-byte lines of one row segment are given to workgroups Q consecutive "slots" apart on the SAME
    // XCD (workgroup b runs on XCD b % 8; an observed placement, used for speed only -- any placement gives the same result),
    // so the second half of every line is served by that XCD's L2 instead of being fetched from memory again by another
    // XCD, and the two half-line stores meet in one L2.  A permutation of [0, full); tiles beyond keep their index.
    constexpr int Q = (CB > 1 && G == 1 && CB * sizeof(V) < 128) ? (int)(128 / (CB * sizeof(V))) : 1;
    const unsigned remap_full = (Q > 1 && tiles_per_a % Q == 0) ? (ntiles / (8u * Q)) * (8u * Q) : 0u;
    auto map_tile = [&](unsigned lin) -> unsigned {
        if constexpr (Q > 1) {
            if (lin < remap_full) {
                const unsigned xcd = lin & 7u, s = lin >> 3;
                return ((s / Q) * 8u + xcd) * Q + (s % Q);
            }
        }
        return lin;
    };
    // point rotation: plane of point k of this thread = j + T * k, so its rotation is (rot_j + k * rot_t) mod row length
    const int rot_j = Tune::ROT ? (rm.rot * j) & rm.mask : 0, rot_t = Tune::ROT ? (rm.rot * T) & rm.mask : 0;
    // loads the E points of the tile group starting at t into dst (zeros for tiles / columns past the end)
    // (points [K0, K1) of the thread only: the partial prefetch below)
    auto load_part = [&](unsigned t, V* dst, auto k0c, auto k1c) {
        constexpr int K0 = decltype(k0c)::value, K1 = decltype(k1c)::value;
        const unsigned tile = map_tile(t + g);
        bool ok = tile < ntiles;
        const unsigned al = tile / tiles_per_a;
        const unsigned b = tile - al * tiles_per_a;
        const unsigned a = al + a_first;
        if (GENERAL) ok = ok && ((int)(b * CB) + c < ncols);
        // rotated rows (exchange buffers): the tile's first column moves inside its row -- by the tile's plane (mode 1) or,
        // per point, by the plane the point belongs to (mode 2); the sides this applies to have unit column stride
        int cb0 = (int)(b * CB);
        if constexpr (Tune::ROT_IN == 1) cb0 = (cb0 + rm.rot * (int)(a + (unsigned)rm.a0)) & rm.mask;
        const GV* ip = in + (long long)a * itile.a_stride + (long long)cb0 * itile.b_stride;
        // (whole-tile prefetch: the per-point offsets irel[0] + k * istep and rotations rot_j + k * rot_t do not depend on the tile,
        // and hoisted out of the tile loop -- which the compiler does -- they cost the very registers PLAIN_IN is there to free: the
        // empty asm statements make them this iteration's values, computed next to the loads that use them)
        int      rj = rot_j;
        unsigned ir0 = irel[0];
        if constexpr (Tune::ROT_IN == 2 && Tune::PLAIN_IN) asm volatile("" : "+v"(rj));
        if constexpr (PLAIN_IN) asm volatile("" : "+v"(ir0));
        if (ok) {
#pragma unroll
            for (int k = K0; k < K1; ++k) {
                long long off = PLAIN_IN ? (long long)(ir0 + (unsigned)k * istep) : (long long)irel[PLAIN_IN ? 0 : k];
                if (GENERAL) off += ((ilast >> k) & 1u) ? (long long)a * imap.last_delta : 0ll;
                if constexpr (Tune::ROT_IN == 2) off += (long long)(((cb0 + rj + k * rot_t) & rm.mask) - cb0);
                dst[k - K0] = VT::from_g(gload<Tune::NTL>(ip + off));
            }
        } else {
#pragma unroll
            for (int k = K0; k < K1; ++k) dst[k - K0] = VT::zero();
        }
    };
    auto load_tile = [&](unsigned t, V* dst) { load_part(t, dst, std::integral_constant<int, 0>{}, std::integral_constant<int, E>{}); };
    // only where the second register set is cheap: <= 32 VGPRs and blocks that do not need the 128-VGPR budget
    // (round 4) ... and where the second set is free: a workgroup of at most 256 threads is one wave per SIMD, which may use all 512
    // registers of its lanes (the compiler parks what does not fit the 256 architectural ones in AGPRs; loads can target them), and
    // a tile beyond 80 KiB means one workgroup per CU anyway -- the 768-point column kernels (24 points per thread, 96 KiB tiles:
    // config 4's Y axis) have nothing in flight underneath their exchanges.  -DDFFT_WIDE_PREFETCH=1 builds it, =2 adds the early wait
    // (the prefetched tile is waited for before this tile's stores are issued, so that they drain underneath the next tile: what the
    // lazy one-launch YZ stage does).  Measured and NOT adopted (round 4, 362-392 registers, no scratch, bit-identical):
    // profiles/r04/experiments/lib_ab_wide_prefetch.log.
#ifndef DFFT_WIDE_PREFETCH
#define DFFT_WIDE_PREFETCH 0
#endif
    constexpr bool WIDE = DFFT_WIDE_PREFETCH && KG::THREADS <= 256 && KG::LDS_BYTES > 80 * 1024 && !GENERAL;
    constexpr bool EARLY = Tune::EARLY_WAIT || (WIDE && DFFT_WIDE_PREFETCH >= 2);
    // 768 points on 12 points x 64 threads (the library's column plan since round 5): its second register set is 48 VGPRs of 16-byte
    // points -- over the general bound, but these kernels have the room.  -DDFFT_768_PREFETCH=0 compiles it out (A/B builds).
#ifndef DFFT_768_PREFETCH
#define DFFT_768_PREFETCH 1
#endif
    constexpr bool P768 = DFFT_768_PREFETCH && P::N == 768 && E == 12 && sizeof(V) == 16;
    constexpr bool PREFETCH = Tune::PREFETCH && (E * (int)sizeof(V) / 4 <= DFFT_PREFETCH_MAX_REGS || Tune::FULL_PREFETCH || WIDE || P768) && KG::THREADS <= 512;
    // 16 points per thread (1024- and 2048-point columns) without Tune::FULL_PREFETCH: a whole second register set does not fit next
    // to per-point offsets (64 VGPRs: 4.6 -> 3.7 TB/s, round 1), but the kernels leave room for HALF of one -- the first 8 points of
    // the next tile are fetched underneath the current tile's exchanges and stores, the other 8 at the top of the next iteration.
    // DFFT_HALF_PREFETCH=0 compiles it out.
#ifndef DFFT_HALF_PREFETCH
#define DFFT_HALF_PREFETCH 1
#endif
    // (only the staged-store variant -- the X pass -- has the registers: 192 -> 228; the plain column variants sit at 216 and would
    // spill 28-68 B with it.  Measured: 1024-point X pass on fp32 column pairs 4.5 -> 4.9 TB/s, fp64 unchanged at 4.6-4.7;
    // profiles/r03/experiments/half_prefetch_ab.log)
    constexpr int PF = PREFETCH ? E : (DFFT_HALF_PREFETCH && Tune::PREFETCH && KG::OSTAGE && E == 16 && sizeof(V) == 16 && KG::THREADS <= 512 && KG::PH == 1 ? E / 2 : 0);
    constexpr bool PARTIAL = PF > 0 && PF < E;
    using K0 = std::integral_constant<int, 0>;
    using KP = std::integral_constant<int, PF>;
    using KE = std::integral_constant<int, E>;
    V v[E];
    V vnext[PF > 0 ? PF : 1];
    if constexpr (PREFETCH) {
        if (blockIdx.x * G < ntiles) load_tile(blockIdx.x * G, v);
    } else if constexpr (PARTIAL) {
        if (blockIdx.x * G < ntiles) load_part(blockIdx.x * G, v, K0{}, KP{});
    }
    for (unsigned t0 = blockIdx.x * G; t0 < ntiles; t0 += tstep) {
        const unsigned tile = map_tile(t0 + g);
        bool valid = tile < ntiles;
        const unsigned al = tile / tiles_per_a;
        const unsigned b = tile - al * tiles_per_a;
        const unsigned a = al + a_first;  // launches over a sub-range of `a` (plane chunks) keep global addressing
        if (GENERAL) valid = valid && ((int)(b * CB) + c < ncols);
        int ob0 = (int)(b * CB);
        if constexpr (Tune::ROT_OUT == 1) ob0 = (ob0 + rm.rot * (int)(a + (unsigned)rm.a0)) & rm.mask;
        GV* op = out + (long long)a * otile.a_stride + (long long)ob0 * otile.b_stride;

        if constexpr (PREFETCH) {
            // issue the next tile's HBM loads now; they complete underneath this tile's exchanges and stores
            if (t0 + tstep < ntiles) load_tile(t0 + tstep, vnext);
        } else if constexpr (PARTIAL) {
            load_part(t0, v + PF, KP{}, KE{});                                   // the rest of this tile
            if (t0 + tstep < ntiles) load_part(t0 + tstep, vnext, K0{}, KP{});  // the first points of the next one
        } else {
            load_tile(t0, v);
        }

        // (729-point column tiles with the ragged / uneven-slab address terms compiled in: 12 bytes over with the thread-local exchange)
        constexpr bool LOCALX = KG::LOCALX && (DFFT_729_LOCALX || !(GENERAL && P::N == 729 && CB > 1));
        run_stages<V, P, 0, DIR, CB, KG::PAD, KG::WAVE_LOCAL, KG::TWMODE, TWPOW, KG::PH, 1, KG::NW, LOCALX>(v, twr, lds, j, c);

        // normalisation folded into this pass: every result is multiplied on its way out (x * 1.0 is exact, so the default
        // changes nothing; a branch around a separate scaling loop cost 15 VGPRs and made the 16-point kernels spill)
        const typename real_of<W>::type sc = (typename real_of<W>::type)scale;

        if constexpr (KG::OSTAGE) {
            // results (column c, idx = j + T*k) -> LDS [scalar column][N + OPAD] -> linear order, so a wave stores
            // contiguous runs (1 KiB per instruction) instead of CB separate 128-byte segments
            static_assert(!GENERAL || !KG::OSTAGE, "the staged store is a fast-path variant");
            constexpr int ROW = N + KG::OPAD;
            W* img = reinterpret_cast<W*>(lds);
            if constexpr (KG::PH == 1) {
                group_sync<KG::WAVE_LOCAL>();
#pragma unroll
                for (int k = 0; k < E; ++k)
#pragma unroll
                    for (int l = 0; l < LANES; ++l) img[(c * LANES + l) * ROW + j + T * k] = VT::lane(cscale(v[k], sc), l);
                group_sync<KG::WAVE_LOCAL>();
                if constexpr (EARLY && PF > 0) {
#pragma unroll
                    for (int k = 0; k < PF; ++k) pin_loaded(vnext[k]);
                }
                if (valid) {
#pragma unroll
                    for (int k = 0; k < E; ++k) {
                        const int lin = tid + GT * k;
                        const GV r = *reinterpret_cast<const GV*>(img + (lin / ON) * ROW + (lin % ON) * LANES);
                        gstore<Tune::NTS>(op + ostage_off(k), r);
                    }
                }
            } else {
                // two images of CB/2 columns each through the same buffer; every thread stores E/2 elements per image
                constexpr int CH = CB / KG::PH, EPH = E / KG::PH;
                const int     mine = c / CH, cl = c - mine * CH;
#pragma unroll
                for (int ph = 0; ph < KG::PH; ++ph) {
                    __syncthreads();
                    if (mine == ph) {
#pragma unroll
                        for (int k = 0; k < E; ++k)
#pragma unroll
                            for (int l = 0; l < LANES; ++l) img[(cl * LANES + l) * ROW + j + T * k] = VT::lane(cscale(v[k], sc), l);
                    }
                    __syncthreads();
                    if (valid) {
#pragma unroll
                        for (int k = 0; k < EPH; ++k) {
                            const int lin = tid + GT * k;
                            const GV r = *reinterpret_cast<const GV*>(img + (lin / ON) * ROW + (lin % ON) * LANES);
                            gstore<Tune::NTS>(op + ostage_off(ph * EPH + k), r);
                        }
                    }
                }
            }
        } else if (valid) {
            if constexpr (EARLY && PF > 0) {
#pragma unroll
                for (int k = 0; k < PF; ++k) pin_loaded(vnext[k]);
            }
            unsigned or0 = orel[0];
            if constexpr (PLAIN) asm volatile("" : "+v"(or0));  // per tile, not a loop invariant (see load_part)
#pragma unroll
            for (int k = 0; k < E; ++k) {
                long long off = PLAIN ? (long long)(or0 + (unsigned)k * ostep) : (long long)orel[PLAIN ? 0 : k];
                if (GENERAL) off += ((olast >> k) & 1u) ? (long long)a * omap.last_delta : 0ll;
                if constexpr (Tune::ROT_OUT == 2) off += (long long)(((ob0 + rot_j + k * rot_t) & rm.mask) - ob0);
                gstore<Tune::NTS>(op + off, VT::to_g(cscale(v[k], sc)));
            }
        }
        if constexpr (PF > 0) {
#pragma unroll
            for (int k = 0; k < PF; ++k) v[k] = vnext[k];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Dual half-line tiles.  A length whose full-line column tile (8 x 16 B) does not fit the LDS (2048 points: 256 KiB) runs
// on 4-column tiles; fft_tiles_kernel then reads and writes 64-byte half lines.  Here a workgroup owns the TWO tiles that
// share the 128-byte lines of a row segment: every thread loads and stores the element of column c AND of column c + CB
// (a wave touches whole lines), and the two 4-column transforms go through the one LDS tile one after the other while the
// other half waits in registers.  Fast path of the transposing store only (forward X pass; no ragged tile, no uneven slab),
// staged per half through the same LDS; everything else stays on fft_tiles_kernel.
// WPC: workgroups meant to be resident per CU (each gets 160 KiB / WPC of LDS and 512 / WPC registers per lane)
template <class V, class P, int CB, int WPC = 1> struct DualGeom {
    using W = typename VecTraits<V>::W;
    static constexpr int    LANES = VecTraits<V>::LANES, OPAD = LANES == 2 ? 2 : 1;
    static constexpr size_t BUDGET = 160 * 1024 / WPC;
    static constexpr size_t TW_BYTES = ((size_t)P::N * sizeof(W) + 15) / 16 * 16;
    static constexpr bool   PADROW = (size_t)(P::N + OPAD) * CB * sizeof(V) + TW_BYTES <= BUDGET;
    static constexpr int    ROW = PADROW ? P::N + OPAD : P::N;  // in units of W; CB * LANES rows
    static constexpr size_t LDS_BYTES = TW_BYTES + (size_t)ROW * CB * sizeof(V);
    static_assert(LDS_BYTES <= BUDGET, "dual tiles: tile + twiddle table must fit the workgroup's share of the CU's LDS");
};
template <class V, class P, int CB, int DIR, bool NT, bool ROT = false, int WPC = 1>
__global__ void __attribute__((amdgpu_flat_work_group_size(1, CB * P::T), amdgpu_waves_per_eu(WPC)))
fft_dual_tiles_kernel(const typename VecTraits<V>::G* in, typename VecTraits<V>::G* out, const typename VecTraits<V>::W* __restrict__ tw,
                      AxisMap imap, AxisMap omap, TileMap itile, TileMap otile, unsigned ntiles, unsigned tiles_per_a, unsigned a_first,
                      double scale, RotMap rm) {
    using VT = VecTraits<V>;
    using W = typename VT::W;
    using GV = typename VT::G;
    constexpr int E = P::E, T = P::T, GT = CB * T, N = P::N, LANES = VT::LANES;
    static_assert(P::S > 1 && GT <= 1024, "dual tiles: multi-stage plans, one workgroup per tile pair");
    // Staged image of the transposing store, one row per scalar column.  Where the LDS has room next to the twiddle table the
    // rows are padded (N + OPAD elements: affine addresses, immediate offsets); at N = 2048 in fp64 tile + table are exactly
    // 160 KiB, so the rows stay N long and row s is rotated by ROT * s elements instead.  Measured
    // (profiles/r02/experiments/dual_tiles.log): any rotation that is not a multiple of 8 memory elements gives 4.8-4.9 TB/s,
    // multiples of 8 (8, 16) 4.5.
    using DG = DualGeom<V, P, CB, WPC>;
    constexpr bool PADROW = DG::PADROW;
    constexpr int  ROW = DG::ROW, IMGROT = LANES;
    auto img_at = [](int col, int e) -> int {  // element e of scalar column col, in units of W
        if constexpr (PADROW) return col * ROW + e;
        else return col * N + ((e + IMGROT * col) & (N - 1));
    };
    constexpr size_t TW_BYTES = ((size_t)N * sizeof(W) + 15) / 16 * 16;
    extern __shared__ __attribute__((aligned(16))) char dfft_smem[];
    W* ldstw = reinterpret_cast<W*>(dfft_smem);
    V* lds = reinterpret_cast<V*>(dfft_smem + TW_BYTES);
    constexpr int NWV = owned_waves<V, CB, T, 1, LANES == 2>();  // (half-line tiles: 0 -- see owned_waves)
    const int tid = threadIdx.x, c = tid % CB, j = tile_j<CB, NWV>(tid);
#if DFFT_TW_STAGE_MAJOR
    fill_stage_major<W, P, 0, DIR, NWV>(ldstw, tw, tid, GT);
#else
    for (int i = tid; i < N; i += GT) {
        W w = tw[i];
        if (DIR < 0) w.y = -w.y;
        ldstw[i] = w;
    }
#endif
    __syncthreads();
    // The launcher guarantees imap.blk % T == 0: the block a point j + T*k falls into depends on k alone, so its
    // offset splits into a wave-uniform term per k and ONE per-thread term -- no per-point address registers next to the two
    // register sets.  (Writing the accesses as uniform base + 32-bit per-thread byte offset did not make the compiler pick the
    // SGPR-base form of the global instructions; it cost 7-41 spilled registers instead.)
    long long iuni[E];
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int ib = (T * k) / imap.blk;
        iuni[k] = block_term(imap, ib) + (long long)(T * k - ib * imap.blk) * imap.stride;
    }
    const typename real_of<W>::type sc = (typename real_of<W>::type)scale;
    constexpr int ON = N / LANES;  // memory elements per scalar column of the staged image
    static_assert((N & (N - 1)) == 0 && (ON % GT == 0 || GT % ON == 0), "staged store: power-of-two geometry");
    // Variants measured and dropped (profiles/r02/experiments/dual_tiles.log, X pass of 2048 x 1024 x 512 fp64: this 4.8 TB/s,
    // one 4-column tile per workgroup 4.0): half 1 loaded underneath the transform of half 0 -- 3.9, the other half line has
    // left the L2 by then; half 1 on the point index j ^ 1 so that a wave instruction fetches whole lines, values sorted into
    // their halves by selects -- 4.4 with 21 spilled registers.  Plain (non-transposing) column passes gain nothing from
    // pairing (3.3 vs 3.6 TB/s: the XCD-aware tile order of fft_tiles_kernel already brings the half lines together in one L2
    // and that kernel keeps two workgroups' worth of loads in flight), so launch_plan pairs tiles for the transposing store only.
    const long long ithr = (long long)j * imap.stride + (long long)c * imap.cstride;
    // rotated rows of the receive buffer (RotMap mode 2): the pair of tiles of plane j + T k starts rot * plane further on in its row
    const int rot_j = ROT ? (rm.rot * j) & rm.mask : 0, rot_t = ROT ? (rm.rot * T) & rm.mask : 0;
    // -DDFFT_DUAL_PIPELINE=1 (build-time experiment of round 4, measured and NOT adopted): a software pipeline over tile pairs.
    // The two register sets hold the whole pair (2 E points of 16 bytes per thread: nothing is left for a third set), so the
    // shipped form loads a pair, transforms and stores its halves, and only then loads the next pair -- with nothing in flight
    // underneath the arithmetic.  The pipelined form refills the registers of a half as soon as its stores have been issued (below).
    // Bit-identical, no scratch in fp64 (242 VGPRs), 8-16 B on fp32 pairs -- and slower in 6 of 8 plans, two processes each,
    // interleaved with the shipped build (profiles/r04/experiments/lib_ab_pipelined_2048.log): X pass of 2048 x 1024 x 512 fp64 7.55 ->
    // 8.11 ms, config 5's per rank (fp32, rotated rows, P = 8) 1.86 -> 2.07, P = 4 3.71 -> 4.1; only the un-rotated fp32 form gained
    // (4.58 -> 4.16).  Every iteration still ends in a full drain (the second refill group is needed at the top of the next one),
    // the refills are half as deep as the whole-pair burst, and the body is written out twice (19 -> 38 KiB of code).
#ifndef DFFT_DUAL_PIPELINE
#define DFFT_DUAL_PIPELINE 0
#endif
#if DFFT_DUAL_PIPELINE
    auto in_ptr = [&](unsigned t) -> const GV* {
        const unsigned al = t / tiles_per_a, b = t - al * tiles_per_a, a = al + a_first;
        return in + (long long)a * itile.a_stride + (long long)b * (2 * CB) * itile.b_stride + ithr;
    };
    auto out_ptr = [&](unsigned t, int h) -> GV* {
        const unsigned al = t / tiles_per_a, b = t - al * tiles_per_a, a = al + a_first;
        return out + (long long)a * otile.a_stride + (long long)b * (2 * CB) * otile.b_stride + (long long)h * CB * LANES * omap.cstride;  // omap.cstride: distance between SCALAR columns
    };
    // One half of a tile pair: the 4-column transform through the LDS tile, the staged image, the transposing store.
    auto process_half = [&](V* v, GV* oh) {
        run_stages<V, P, 0, DIR, CB, false, false, TW_LDS, false, 1, 1, NWV>(v, ldstw, lds, j, c);
        W* img = reinterpret_cast<W*>(lds);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < E; ++k)
#pragma unroll
            for (int l = 0; l < LANES; ++l) {
                const int col = c * LANES + l;
                img[img_at(col, j + T * k)] = VT::lane(cscale(v[k], sc), l);
            }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < E; ++k) {
            const int lin = tid + GT * k;
            const int col = lin / ON;
            const GV  r = *reinterpret_cast<const GV*>(img + img_at(col, (lin % ON) * LANES));
            gstore<NT>(oh + (long long)col * omap.cstride + (lin % ON), r);
        }
        __syncthreads();  // the image is read before the next half's exchange overwrites the tile
    };
    // point k of both halves of tile pair t (the two loads of a lane group are the two halves of the same 128-byte lines)
    auto load_point = [&](unsigned t, int k, V& a0, V& a1) {
        const GV* ip = in_ptr(t);
        long long off = iuni[k];
        if constexpr (ROT) {
            const int cb0 = (int)((t % tiles_per_a) * (2 * CB));
            off += (long long)(((cb0 + rot_j + k * rot_t) & rm.mask) - cb0);
        }
        a0 = VT::from_g(gload<NT>(ip + off));
        a1 = VT::from_g(gload<NT>(ip + off + (long long)CB * imap.cstride));
    };
    // The registers of a half are refilled as soon as that half's stores have been issued: after half 0
    // the points k < E/2 of BOTH halves of the next pair (a lane group still fetches whole 128-byte lines -- loading one half's
    // columns first would fetch every line twice, profiles/r02/experiments/dual_tiles.log), after half 1 the points k >= E/2.
    // The loads of the first group complete underneath the second half's exchanges, those of the second group underneath
    // the drain of its stores.  Loaded registers must not be moved (a move waits for its load), so the assignment of points to
    // registers alternates between two layouts and the loop body is written out for both:
    //   layout 0: A = half 0 (all k), B = half 1;   layout 1: half 0 = {A[k], B[k]}, half 1 = {A[E/2 + k], B[E/2 + k]}, k < E/2
    constexpr int EH = E / 2;
    static_assert(E % 2 == 0, "dual tiles: even number of points per thread");
    V A[E], B[E];
    unsigned t = blockIdx.x;
    if (t < ntiles) {
#pragma unroll
        for (int k = 0; k < E; ++k) load_point(t, k, A[k], B[k]);
    }
    // (the prefetch is unconditional: past the workgroup's last pair it re-reads that pair -- two branches less in a loop body
    // that sits at the register limit)
    const unsigned step = gridDim.x;
    while (t < ntiles) {
        {  // layout 0 -> 1
            const unsigned tn = t + step, tl = tn < ntiles ? tn : t;
            process_half(A, out_ptr(t, 0));
#pragma unroll
            for (int k = 0; k < EH; ++k) load_point(tl, k, A[k], A[EH + k]);
            process_half(B, out_ptr(t, 1));
#pragma unroll
            for (int k = 0; k < EH; ++k) load_point(tl, EH + k, B[k], B[EH + k]);
            t = tn;
            if (t >= ntiles) break;
        }
        {  // layout 1 -> 0
            const unsigned tn = t + step, tl = tn < ntiles ? tn : t;
            V              w0[E], w1[E];
#pragma unroll
            for (int k = 0; k < EH; ++k) {
                w0[k] = A[k];
                w0[EH + k] = B[k];
                w1[k] = A[EH + k];
                w1[EH + k] = B[EH + k];
            }
            process_half(w0, out_ptr(t, 0));
#pragma unroll
            for (int k = 0; k < EH; ++k) load_point(tl, k, A[k], B[k]);
            process_half(w1, out_ptr(t, 1));
#pragma unroll
            for (int k = 0; k < EH; ++k) load_point(tl, EH + k, A[EH + k], B[EH + k]);
            t = tn;
        }
    }
#else
    for (unsigned t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const unsigned al = t / tiles_per_a, b = t - al * tiles_per_a, a = al + a_first;
        const GV*      ip = in + (long long)a * itile.a_stride + (long long)b * (2 * CB) * itile.b_stride + ithr;
        GV*            op = out + (long long)a * otile.a_stride + (long long)b * (2 * CB) * otile.b_stride;
        V v0[E], v1[E];
#pragma unroll
        for (int k = 0; k < E; ++k) {
            long long off = iuni[k];
            if constexpr (ROT) {
                const int cb0 = (int)(b * (2 * CB));
                off += (long long)(((cb0 + rot_j + k * rot_t) & rm.mask) - cb0);
            }
            v0[k] = VT::from_g(gload<NT>(ip + off));
            v1[k] = VT::from_g(gload<NT>(ip + off + (long long)CB * imap.cstride));
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            V* v = h == 0 ? v0 : v1;
            run_stages<V, P, 0, DIR, CB, false, false, TW_LDS, false, 1, 1, NWV>(v, ldstw, lds, j, c);
            W* img = reinterpret_cast<W*>(lds);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < E; ++k)
#pragma unroll
                for (int l = 0; l < LANES; ++l) {
                    const int col = c * LANES + l;
                    img[img_at(col, j + T * k)] = VT::lane(cscale(v[k], sc), l);
                }
            __syncthreads();
            GV* oh = op + (long long)h * CB * LANES * omap.cstride;  // omap.cstride: distance between SCALAR columns
#pragma unroll
            for (int k = 0; k < E; ++k) {
                const int lin = tid + GT * k;
                const int col = lin / ON;
                const GV  r = *reinterpret_cast<const GV*>(img + img_at(col, (lin % ON) * LANES));
                gstore<NT>(oh + (long long)col * omap.cstride + (lin % ON), r);
            }
            __syncthreads();  // the image is read before the next half's exchange overwrites the tile
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Full-line column tiles for a length N = 2 * NH whose own full-line tile does not fit the LDS (2048 points: 256 KiB) but whose
// half does: the first radix-2 stage is done decimation-in-frequency AT LOAD TIME.  Thread j of a column holds the points
// n = j + T k and n + NH (k < E; PH = the NH-point plan, E T = NH) -- exactly the two inputs of every first-stage butterfly --
//     a[n] = x[n] + x[n + NH]                  -> even outputs  X[2m]     = DFT_NH(a)[m]
//     b[n] = (x[n] - x[n + NH]) * W_N^n        -> odd  outputs  X[2m + 1] = DFT_NH(b)[m]
// and the two NH-point transforms go through the one (NH x CB) LDS tile one after the other while the other half waits in
// registers.  Every HBM access is a full 128-byte line on both sides (the half-line kernel of fft_tiles_kernel moves 64-byte
// segments), with 2 E loads per thread in flight.  W_N^{j + T k} = W_N^j * W_N^{T k}: one per-thread factor times a
// wave-uniform one, so the stage needs no table of its own; the NH-point stages read the even entries of the N-point table
// from an LDS copy.  Fast path only (no ragged tile, no uneven slab, natural or packed maps whose blocks are multiples of
// T / 2 T points); the transposing store of the forward X pass stays on fft_dual_tiles_kernel.
// TOUT (round 6): the OUTPUT side is the transposed one ([..][z][kx], kx fastest: the forward X pass).  The two half transforms leave
// X[2m] and X[2m + 1] -- neighbours in kx -- in the registers of one thread; they are staged through an LDS image of [scalar column][kx]
// rows, CB / 2 columns at a time (the image of all CB columns of an N-point tile is twice the LDS), and stored in linear order, 1 KiB
// runs per wave instruction, like the staged store of fft_tiles_kernel.  Against the paired half-line tiles of fft_dual_tiles_kernel
// (the 2048-point X pass until round 6): full 128-byte lines on the load side without pairing tiles, two workgroup barriers per half
// transform instead of six (the second exchange is wave-owned, the third thread-local), 8 per 256 KiB of data instead of 16.
template <class V, class PH, int CB, bool TOUT = false> struct Dif2Geom {
    using W = typename VecTraits<V>::W;
    static constexpr int    LANES = VecTraits<V>::LANES, OPAD = LANES == 2 ? 2 : 1, N = 2 * PH::N;
    static constexpr size_t TW_BYTES = ((size_t)2 * PH::N * sizeof(W) + 15) / 16 * 16;  // the whole N-point table
    static constexpr size_t TILE_BYTES = (size_t)PH::N * CB * sizeof(V);
    // image of CB / 2 columns: padded rows where they fit next to the table, else rows of N rotated by their index (fft_dual_tiles_kernel)
    // (... and where rows of exactly N let TWO workgroups share the CU -- 1024 fp64 points: 16 + 64 KiB -- the pad is not worth the second
    // workgroup it would cost)
    static constexpr size_t IMG_PLAIN = (size_t)(CB / 2) * LANES * N * sizeof(W);
    static constexpr bool   TWO_PER_CU = TOUT && TW_BYTES + (TILE_BYTES > IMG_PLAIN ? TILE_BYTES : IMG_PLAIN) <= 80 * 1024;
    static constexpr size_t BUDGET = TWO_PER_CU ? 80 * 1024 : 160 * 1024;
    // (-DDFFT_DIF2_HALF experiment) half-line tiles of the non-transposing passes: two workgroups per CU where half tile + table allow it
    static constexpr bool   HALF2 = !TOUT && (size_t)CB * sizeof(V) < 128 && TW_BYTES + TILE_BYTES <= 80 * 1024;
    static constexpr bool   PADROW = TW_BYTES + (size_t)(CB / 2) * LANES * (N + OPAD) * sizeof(W) <= BUDGET;
    static constexpr int    ROW = PADROW ? N + OPAD : N;
    static constexpr size_t IMG_BYTES = TOUT ? (size_t)(CB / 2) * LANES * ROW * sizeof(W) : 0;
    static constexpr size_t LDS_BYTES = TW_BYTES + (TILE_BYTES > IMG_BYTES ? TILE_BYTES : IMG_BYTES);
    static_assert(LDS_BYTES <= 160 * 1024, "DIF-split tiles: half tile + twiddle table must fit the CU's LDS");
};
// BIN / BOUT: the side's wave-uniform offsets come from a table computed once (any map); false = k * step for single-block maps,
// which measured 150 B of scratch against none with the tables, so the launcher always asks for both.
template <class V, class PH, int CB, int DIR, bool NTL, bool NTS, bool BIN, bool BOUT, int ROT = 0, bool TOUT = false>
__global__ void __attribute__((amdgpu_flat_work_group_size(1, CB * PH::T), amdgpu_waves_per_eu((Dif2Geom<V, PH, CB, TOUT>::TWO_PER_CU || Dif2Geom<V, PH, CB, TOUT>::HALF2) ? (CB * PH::T) / 128 : 1)))
fft_dif2_tiles_kernel(const typename VecTraits<V>::G* in, typename VecTraits<V>::G* out, const typename VecTraits<V>::W* __restrict__ tw,
                      AxisMap imap, AxisMap omap, TileMap itile, TileMap otile, unsigned ntiles, unsigned tiles_per_a, unsigned a_first,
                      double scale, RotMap rm) {
    using VT = VecTraits<V>;
    using W = typename VT::W;
    using GV = typename VT::G;
    constexpr int E = PH::E, T = PH::T, GT = CB * T, NH = PH::N, N = 2 * NH, LANES = VT::LANES;
    static_assert(PH::S > 1 && E * T == NH && GT <= 1024, "DIF-split tiles: multi-stage half plan, one workgroup per tile");
    static_assert(!TOUT || (BOUT == false && (ROT == 0 || ROT == 2) && CB % 2 == 0 && ((CB / 2) * (N / LANES)) % GT == 0 && (N / LANES) % GT == 0),
                  "staged transposed output: rotation per point on the input side only, image phases that split evenly over the workgroup");
    using DG = Dif2Geom<V, PH, CB, TOUT>;
    extern __shared__ __attribute__((aligned(16))) char dfft_smem[];
    W* ldstw = reinterpret_cast<W*>(dfft_smem);  // N entries; the NH-point stages use every second one
    V* lds = reinterpret_cast<V*>(dfft_smem + DG::TW_BYTES);
    constexpr int NWV = owned_waves<V, CB, T>();
    const int tid = threadIdx.x, c = tid % CB, j = tile_j<CB, NWV>(tid);
#if DFFT_TW_STAGE_MAJOR
    // (round 6) the same N entries of LDS, laid out for the lanes that read them: [0, NH) the stage-major table of the NH-point stages
    // (every second entry of the N-point table, rows in lane order: tw_row), [NH, N) the first-stage factors W_N^n, n = j + T k, as
    // [k][ids in lane order] -- under the wave-interleaved labelling the lanes of a wavefront hold ids NW apart, and read from the
    // natural-order table they would all hit the same banks
    constexpr int TWS_RUN = 1;
    W*            ldsf = ldstw + NH;
    fill_stage_major<W, PH, 0, DIR, NWV, 2>(ldstw, tw, tid, GT);
    auto fperm = [](int jj) -> int {
        if constexpr (NWV > 1) return (jj % NWV) * (T / NWV) + jj / NWV;
        else return jj;
    };
    for (int i = tid; i < NH; i += GT) {
        W w = tw[i];
        if (DIR < 0) w.y = -w.y;
        ldsf[(i / T) * T + fperm(i % T)] = w;
    }
    const W* fst = ldsf + fperm(j);  // factor of point k: fst[T k]
#else
    constexpr int TWS_RUN = 2;
    for (int i = tid; i < N; i += GT) {
        W w = tw[i];
        if (DIR < 0) w.y = -w.y;
        ldstw[i] = w;
    }
    const W* fst = ldstw + j;
#endif
    __syncthreads();
    // Multi-block sides: the launcher guarantees imap.blk % T == 0 and omap.blk % (2 T) == 0, so the block a point falls into
    // depends on k alone -- a wave-uniform 32-bit term per k (computed once) plus ONE per-thread term.  Plain sides: k * step.
    unsigned iuni[BIN ? 2 * E : 1], ouni[BOUT ? E : 1];
    if constexpr (BIN) {
#pragma unroll
        for (int kk = 0; kk < 2 * E; ++kk) {
            const int ib = (T * kk) / imap.blk;
            iuni[kk] = (unsigned)(block_term(imap, ib) + (long long)(T * kk - ib * imap.blk) * imap.stride);
        }
    }
    if constexpr (BOUT) {
#pragma unroll
        for (int k = 0; k < E; ++k) {
            const int ob = (2 * T * k) / omap.blk;
            ouni[k] = (unsigned)(block_term(omap, ob) + (long long)(2 * T * k - ob * omap.blk) * omap.stride);
        }
    }
    const long long istep = (long long)T * imap.stride, ostep = (long long)(2 * T) * omap.stride;
    auto in_off = [&](int kk) -> long long {  // points j + T kk, kk < 2 E
        if constexpr (BIN) return (long long)iuni[kk];
        else return kk * istep;
    };
    auto out_off = [&](int k) -> long long {  // output rows 2 j + h + 2 T k
        if constexpr (BOUT) return (long long)ouni[k];
        else return k * ostep;
    };
    const long long ithr = (long long)j * imap.stride + (long long)c * imap.cstride;
    const long long othr = (long long)(2 * j) * omap.stride + (long long)c * omap.cstride;
    const typename real_of<W>::type sc = (typename real_of<W>::type)scale;
    // -DDFFT_DIF2_PIPELINE=1 (build-time experiment of round 4, measured and NOT adopted; see fft_dual_tiles_kernel): t0 of
    // 512 x 2048 x 512 fp64 6.19 -> 5.97 ms, but fp32 3.16 -> 3.66, config 5's per-rank t0 3.27 -> 3.37 (12 B of scratch on fp32 pairs).
#ifndef DFFT_DIF2_PIPELINE
#define DFFT_DIF2_PIPELINE 0
#endif
#if DFFT_DIF2_PIPELINE
    auto in_ptr = [&](unsigned t) -> const GV* {
        const unsigned al = t / tiles_per_a, b = t - al * tiles_per_a, a = al + a_first;
        // rotated rows of an exchange buffer (RotMap mode 1): the whole tile moves inside its row by the plane's rotation
        int cbi = (int)(b * CB);
        if constexpr (ROT == 1) {
            if (rm.in_mode == 1) cbi = (cbi + rm.rot * (int)(a + (unsigned)rm.a0)) & rm.mask;
        }
        return in + (long long)a * itile.a_stride + (long long)cbi * itile.b_stride + ithr;
    };
    auto out_ptr = [&](unsigned t) -> GV* {
        const unsigned al = t / tiles_per_a, b = t - al * tiles_per_a, a = al + a_first;
        int cbo = (int)(b * CB);
        if constexpr (ROT == 1) {
            if (rm.out_mode == 1) cbo = (cbo + rm.rot * (int)(a + (unsigned)rm.a0)) & rm.mask;
        }
        return out + (long long)a * otile.a_stride + (long long)cbo * otile.b_stride + othr;
    };
    // the two inputs n = j + T k and n + NH of first-stage butterfly k
    auto load_pair = [&](unsigned t, int k, V& x0, V& x1) {
        const GV* ip = in_ptr(t);
        x0 = VT::from_g(gload<NTL>(ip + in_off(k)));
        x1 = VT::from_g(gload<NTL>(ip + in_off(k + E)));
    };
    auto split = [&](V& x0, V& x1, int k) {  // (x[n], x[n + NH]) -> (a[n], b[n])
        const V sum = cadd(x0, x1);
        const V dif = csub(x0, x1);
        x0 = sum;
        x1 = cmul(dif, fst[T * k]);  // W_N^{j + T k}
    };
    auto process_half = [&](V* v, unsigned t, int h) {
        run_stages<V, PH, 0, DIR, CB, false, false, TW_LDS, false, 1, TWS_RUN, NWV>(v, ldstw, lds, j, c);
        GV* op = out_ptr(t);
#pragma unroll
        for (int k = 0; k < E; ++k) gstore<NTS>(op + out_off(k) + (long long)h * omap.stride, VT::to_g(cscale(v[k], sc)));
    };
    // Software pipeline over tiles, as in fft_dual_tiles_kernel: the registers of a half transform are refilled as soon as
    // its stores have been issued -- after the even half the input pairs k < E/2 of the next tile, after the odd half the pairs
    // k >= E/2 -- so that half of the next tile's loads complete underneath the odd half's exchanges.  Loaded registers are not
    // moved; the register layout alternates between
    //   layout 0: pair k = (A[k], B[k]);   layout 1: pair k = (A[k], A[E/2 + k]) for k < E/2, (B[k - E/2], B[k]) for k >= E/2
    // and the first-stage butterfly leaves a[k] in the pair's first register, b[k] in its second.
    constexpr int EH = E / 2;
    static_assert(E % 2 == 0, "DIF-split tiles: even number of points per thread");
    V A[E], B[E];
    unsigned t = blockIdx.x;
    if (t < ntiles) {
#pragma unroll
        for (int k = 0; k < E; ++k) load_pair(t, k, A[k], B[k]);
    }
    const unsigned step = gridDim.x;
    while (t < ntiles) {
        {  // layout 0 -> 1
            const unsigned tn = t + step, tl = tn < ntiles ? tn : t;  // (past the last tile: re-read it, see fft_dual_tiles_kernel)
#pragma unroll
            for (int k = 0; k < E; ++k) split(A[k], B[k], k);
            process_half(A, t, 0);
#pragma unroll
            for (int k = 0; k < EH; ++k) load_pair(tl, k, A[k], A[EH + k]);
            process_half(B, t, 1);
#pragma unroll
            for (int k = 0; k < EH; ++k) load_pair(tl, EH + k, B[k], B[EH + k]);
            t = tn;
            if (t >= ntiles) break;
        }
        {  // layout 1 -> 0
            const unsigned tn = t + step, tl = tn < ntiles ? tn : t;
            V              w0[E], w1[E];
#pragma unroll
            for (int k = 0; k < EH; ++k) {
                split(A[k], A[EH + k], k);
                split(B[k], B[EH + k], EH + k);
                w0[k] = A[k];
                w0[EH + k] = B[k];
                w1[k] = A[EH + k];
                w1[EH + k] = B[EH + k];
            }
            process_half(w0, t, 0);
#pragma unroll
            for (int k = 0; k < EH; ++k) load_pair(tl, k, A[k], B[k]);
            process_half(w1, t, 1);
#pragma unroll
            for (int k = 0; k < EH; ++k) load_pair(tl, EH + k, A[EH + k], B[EH + k]);
            t = tn;
        }
    }
#else
    if constexpr (TOUT) {
        // Software pipeline over tiles: the next tile's loads are issued as soon as the second image phase has taken the last results out
        // of the registers -- in front of that phase's stores, so the wait for the loaded points (one vmcnt for loads and stores) lets
        // those stores drain underneath the next tile's first stage instead of in front of it.
        constexpr int CH = CB / 2, ON = N / LANES, ROW = DG::ROW;
        auto img_at = [](int col, int e) -> int {  // element e (units of W) of scalar column col of the phase
            if constexpr (DG::PADROW) return col * ROW + e;
            else return col * N + ((e + LANES * col) & (N - 1));
        };
        W*        img = reinterpret_cast<W*>(lds);
        const int mine = c / CH, cl = c - mine * CH;
        V         v0[E], v1[E];
        // Loads through buffer descriptors: one 32-bit per-thread offset (plus the per-point rotation) and a wave-uniform scalar offset
        // k * step per point instead of a 64-bit address register pair per load -- with 2 E loads in flight next to the image reads of the
        // second phase the flat form went 276-308 bytes over the 256 registers.  Two descriptors, one per half of the points (the slab
        // side of config 5's X pass spans 4 GiB per rank at P = 8, 8 GiB at P = 4; the launcher checks that a half stays below 4 GiB).
        static_assert(!BIN, "staged transposed output: single-block input map (offsets kk * step)");
        const unsigned thr16 = (unsigned)((long long)j * imap.stride + (long long)c * imap.cstride) * 16u;
        auto load_tile = [&](unsigned t) {
            const unsigned al = t / tiles_per_a, b = t - al * tiles_per_a, a = al + a_first;
            const int      cbi = (int)(b * CB);
            const char*    row0 = reinterpret_cast<const char*>(in + (long long)a * itile.a_stride);  // column 0 of the tile's slice
            const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)row0, 0, (int)0xffffffffu, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(row0 + (long long)E * istep * 16), 0, (int)0xffffffffu, 0x00020000);
            int rj = ROT == 2 ? (rm.rot * j) & rm.mask : 0;
            if constexpr (ROT == 2) asm volatile("" : "+v"(rj));  // this tile's value: hoisted, the 2 E per-point rotations would live in registers
            const int rot_t = ROT == 2 ? (rm.rot * T) & rm.mask : 0;
            const unsigned col16 = (unsigned)((long long)cbi * itile.b_stride) * 16u;  // un-rotated: the tile's first column
#pragma unroll
            for (int k = 0; k < E; ++k) {
                unsigned o0 = thr16 + col16, o1 = o0;
                if constexpr (ROT == 2) {  // rotated rows of the receive buffer, per point: plane j + T kk starts rot * plane further on in its row
                    o0 = thr16 + (unsigned)((cbi + rj + k * rot_t) & rm.mask) * 16u;
                    o1 = thr16 + (unsigned)((cbi + rj + (k + E) * rot_t) & rm.mask) * 16u;
                }
                const int so = (int)((unsigned)k * (unsigned)istep * 16u);  // wave-uniform
                v0[k] = VT::from_g(__builtin_bit_cast(GV, __builtin_amdgcn_raw_buffer_load_b128(rs0, (int)o0, so, NTL ? 2 : 0)));
                v1[k] = VT::from_g(__builtin_bit_cast(GV, __builtin_amdgcn_raw_buffer_load_b128(rs1, (int)o1, so, NTL ? 2 : 0)));
            }
        };
        // -DDFFT_DIF2_TOUT_PIPE=1 builds the software pipeline described above: measured and NOT adopted (round 6) -- with the next tile's 2 E
        // points in flight next to the second phase's image reads the kernel needs 270-300 bytes of scratch (flat or buffer loads alike)
#ifndef DFFT_DIF2_TOUT_PIPE
#define DFFT_DIF2_TOUT_PIPE 0
#endif
        unsigned t = blockIdx.x;
        if (DFFT_DIF2_TOUT_PIPE && t < ntiles) load_tile(t);
        for (; t < ntiles; t += gridDim.x) {
            const unsigned al = t / tiles_per_a, b = t - al * tiles_per_a, a = al + a_first;
            if (!DFFT_DIF2_TOUT_PIPE) load_tile(t);
#pragma unroll
            for (int k = 0; k < E; ++k) {
                const V sum = cadd(v0[k], v1[k]);
                const V dif = csub(v0[k], v1[k]);
                v0[k] = sum;
                v1[k] = cmul(dif, fst[T * k]);  // W_N^{j + T k}
            }
            run_stages<V, PH, 0, DIR, CB, false, false, TW_LDS, false, 1, TWS_RUN, NWV>(v0, ldstw, lds, j, c);
            run_stages<V, PH, 0, DIR, CB, false, false, TW_LDS, false, 1, TWS_RUN, NWV>(v1, ldstw, lds, j, c);
            // thread (j, c) now holds X[2 m] in v0[k] and X[2 m + 1] in v1[k], m = j + T k, of column (pair) c: stage them through the
            // image of [scalar column][kx] rows, CH = CB / 2 columns per phase, and store the image in linear order
            GV* ot = out + (long long)a * otile.a_stride + (long long)(b * CB) * otile.b_stride;  // omap.cstride: distance between SCALAR columns
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                __syncthreads();  // the last exchange / the previous phase's image is no longer read
                if (mine == ph) {
#pragma unroll
                    for (int k = 0; k < E; ++k) {
                        const V a0 = cscale(v0[k], sc), a1 = cscale(v1[k], sc);
#pragma unroll
                        for (int l = 0; l < LANES; ++l) {
                            const W e0 = VT::lane(a0, l), e1 = VT::lane(a1, l);
                            if constexpr (LANES == 2 && DG::PADROW) {  // two 8-byte neighbours in kx: one 16-byte write
                                *reinterpret_cast<f32x4*>(img + img_at(cl * LANES + l, 2 * (j + T * k))) = f32x4{e0.x, e0.y, e1.x, e1.y};
                            } else {
                                img[img_at(cl * LANES + l, 2 * (j + T * k))] = e0;
                                img[img_at(cl * LANES + l, 2 * (j + T * k) + 1)] = e1;
                            }
                        }
                    }
                }
                if (DFFT_DIF2_TOUT_PIPE && ph == 1 && t + gridDim.x < ntiles) load_tile(t + gridDim.x);  // every result has left the registers
                __syncthreads();
#pragma unroll
                for (int k = 0; k < E; ++k) {
                    const int lin = tid + GT * k, col = lin / ON, e = lin % ON;
                    const GV  r = *reinterpret_cast<const GV*>(img + img_at(col, e * LANES));
                    gstore<NTS>(ot + (long long)(ph * CH * LANES + col) * omap.cstride + e, r);
                    // four elements at a time: all E image reads in flight at once would need 4 E registers next to the 8 E of the
                    // prefetched tile
                    if (k % 4 == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    } else {
    // half-line tiles (CB columns narrower than a cache line; -DDFFT_DIF2_HALF experiment): the XCD-aware tile order of fft_tiles_kernel --
    // the Q tiles that share the lines of a row segment go to workgroups on the same XCD
    constexpr int  Q = (CB * sizeof(V) < 128) ? (int)(128 / (CB * sizeof(V))) : 1;
    const unsigned remap_full = (Q > 1 && tiles_per_a % Q == 0) ? (ntiles / (8u * Q)) * (8u * Q) : 0u;
    for (unsigned tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        unsigned t = tl;
        if constexpr (Q > 1) {
            if (tl < remap_full) {
                const unsigned xcd = tl & 7u, s = tl >> 3;
                t = ((s / Q) * 8u + xcd) * Q + (s % Q);
            }
        }
        const unsigned al = t / tiles_per_a, b = t - al * tiles_per_a, a = al + a_first;
        // rotated rows of an exchange buffer (RotMap mode 1): the whole tile moves inside its row by the plane's rotation
        int cbi = (int)(b * CB), cbo = (int)(b * CB);
        if constexpr (ROT == 1) {
            const int r = rm.rot * (int)(a + (unsigned)rm.a0);
            if (rm.in_mode == 1) cbi = (cbi + r) & rm.mask;
            if (rm.out_mode == 1) cbo = (cbo + r) & rm.mask;
        }
        const GV*      ip = in + (long long)a * itile.a_stride + (long long)cbi * itile.b_stride + ithr;
        GV*            op = out + (long long)a * otile.a_stride + (long long)cbo * otile.b_stride + othr;
        V v0[E], v1[E];
#pragma unroll
        for (int k = 0; k < E; ++k) {
            v0[k] = VT::from_g(gload<NTL>(ip + in_off(k)));
            v1[k] = VT::from_g(gload<NTL>(ip + in_off(k + E)));
        }
#pragma unroll
        for (int k = 0; k < E; ++k) {
            const V sum = cadd(v0[k], v1[k]);
            const V dif = csub(v0[k], v1[k]);
            v0[k] = sum;
            v1[k] = cmul(dif, fst[T * k]);  // W_N^{j + T k}
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            V* v = h == 0 ? v0 : v1;
            run_stages<V, PH, 0, DIR, CB, false, false, TW_LDS, false, 1, TWS_RUN, NWV>(v, ldstw, lds, j, c);
#pragma unroll
            for (int k = 0; k < E; ++k) gstore<NTS>(op + out_off(k) + (long long)h * omap.stride, VT::to_g(cscale(v[k], sc)));
        }
    }
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Radix-3 split at load time (round 6, last session; -DDFFT_DIF3=1, DFFT_DIF3=1): full-line column tiles for N = 3 * NT whose own tile
// leaves room for ONE workgroup per CU only (768 points: 96 KiB + table).  Thread j of a column holds the points n = j + T k, n + NT and
// n + 2 NT (k < E; PT = the NT-point plan, E T = NT) -- the three inputs of every first-stage butterfly of a decimation-in-frequency
// radix-3 step --
//     u_r[n] = (sum_q x[n + q NT] w3^(q r)) * W_N^(n r)   ->   X[3 m + r] = DFT_NT(u_r)[m],   r = 0, 1, 2
// and the three NT-point transforms go through one (NT x CB) LDS tile one after the other: 768 fp64 points need 32 KiB + 12 KiB of
// tables, so THREE workgroups share a CU (fft_dif2_tiles_kernel's argument for 1024 points, profiles/r06/README.md section 1g), with two
// exchanges per sub-transform instead of three over the whole 96 KiB tile.  W_N^n and W_N^(2 n) (n < NT) come from the N-point table;
// the NT-point stages read every third entry of it, stage-major, from the LDS copy.  Fast path only: whole tiles, both sides keep the
// columns of a line together, input blocks multiples of T points, output blocks multiples of 3 T.
template <class V, class PT, int CB> struct Dif3Geom {
    using W = typename VecTraits<V>::W;
    static constexpr size_t TW_BYTES = ((size_t)3 * PT::N * sizeof(W) + 15) / 16 * 16;  // stage-major table + W^n + W^(2n)
    static constexpr size_t TILE_BYTES = (size_t)PT::N * CB * sizeof(V);
    static constexpr size_t LDS_BYTES = TW_BYTES + TILE_BYTES;
    static constexpr int    PER_CU = (int)(160 * 1024 / LDS_BYTES) < 1 ? 1 : (int)(160 * 1024 / LDS_BYTES);
    static constexpr int    WAVES_PER_EU = (PER_CU * CB * PT::T + 255) / 256;  // waves per SIMD when PER_CU workgroups are resident
};
template <class V, class PT, int CB, int DIR, bool NTL, bool NTS, int ROT = 0>
__global__ void __attribute__((amdgpu_flat_work_group_size(1, CB * PT::T), amdgpu_waves_per_eu(Dif3Geom<V, PT, CB>::WAVES_PER_EU)))
fft_dif3_tiles_kernel(const typename VecTraits<V>::G* in, typename VecTraits<V>::G* out, const typename VecTraits<V>::W* __restrict__ tw,
                      AxisMap imap, AxisMap omap, TileMap itile, TileMap otile, unsigned ntiles, unsigned tiles_per_a, unsigned a_first,
                      double scale, RotMap rm) {
    using VT = VecTraits<V>;
    using W = typename VT::W;
    using GV = typename VT::G;
    constexpr int E = PT::E, T = PT::T, GT = CB * T, NT = PT::N;
    static_assert(PT::S > 1 && E * T == NT && GT <= 1024, "radix-3 split tiles: multi-stage third plan, one workgroup per tile");
    using DG = Dif3Geom<V, PT, CB>;
    extern __shared__ __attribute__((aligned(16))) char dfft_smem[];
    W* ldstw = reinterpret_cast<W*>(dfft_smem);  // [0, NT) stage-major table of the NT-point stages, [NT, 2 NT) W_N^n, [2 NT, 3 NT) W_N^(2 n)
    V* lds = reinterpret_cast<V*>(dfft_smem + DG::TW_BYTES);
    constexpr int NWV = owned_waves<V, CB, T>();
    const int     tid = threadIdx.x, c = tid % CB, j = tile_j<CB, NWV>(tid);
    fill_stage_major<W, PT, 0, DIR, NWV, 3>(ldstw, tw, tid, GT);
    auto fperm = [](int jj) -> int {  // (as in fft_dif2_tiles_kernel: the factors in the order the lanes of a wavefront read them)
        if constexpr (NWV > 1) return (jj % NWV) * (T / NWV) + jj / NWV;
        else return jj;
    };
    for (int i = tid; i < NT; i += GT) {
        W w1 = tw[i], w2 = tw[2 * i];
        if (DIR < 0) {
            w1.y = -w1.y;
            w2.y = -w2.y;
        }
        const int at = (i / T) * T + fperm(i % T);
        ldstw[NT + at] = w1;
        ldstw[2 * NT + at] = w2;
    }
    const W* f1 = ldstw + NT + fperm(j);  // factors of point k: f1[T k], f2[T k]
    const W* f2 = ldstw + 2 * NT + fperm(j);
    __syncthreads();
    unsigned iuni[3 * E], ouni[E];
#pragma unroll
    for (int kk = 0; kk < 3 * E; ++kk) {
        const int ib = (T * kk) / imap.blk;
        iuni[kk] = (unsigned)(block_term(imap, ib) + (long long)(T * kk - ib * imap.blk) * imap.stride);
    }
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int ob = (3 * T * k) / omap.blk;
        ouni[k] = (unsigned)(block_term(omap, ob) + (long long)(3 * T * k - ob * omap.blk) * omap.stride);
    }
    const long long ithr = (long long)j * imap.stride + (long long)c * imap.cstride;
    const long long othr = (long long)(3 * j) * omap.stride + (long long)c * omap.cstride;
    const typename real_of<W>::type sc = (typename real_of<W>::type)scale;
    for (unsigned t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const unsigned al = t / tiles_per_a, b = t - al * tiles_per_a, a = al + a_first;
        int cbi = (int)(b * CB), cbo = (int)(b * CB);
        if constexpr (ROT == 1) {  // rotated rows of an exchange buffer (RotMap mode 1): the whole tile moves inside its row
            const int r = rm.rot * (int)(a + (unsigned)rm.a0);
            if (rm.in_mode == 1) cbi = (cbi + r) & rm.mask;
            if (rm.out_mode == 1) cbo = (cbo + r) & rm.mask;
        }
        const GV* ip = in + (long long)a * itile.a_stride + (long long)cbi * itile.b_stride + ithr;
        GV*       op = out + (long long)a * otile.a_stride + (long long)cbo * otile.b_stride + othr;
        V         v0[E], v1[E], v2[E];
#pragma unroll
        for (int k = 0; k < E; ++k) {
            v0[k] = VT::from_g(gload<NTL>(ip + (long long)iuni[k]));
            v1[k] = VT::from_g(gload<NTL>(ip + (long long)iuni[k + E]));
            v2[k] = VT::from_g(gload<NTL>(ip + (long long)iuni[k + 2 * E]));
        }
#pragma unroll
        for (int k = 0; k < E; ++k) {
            V u[3] = {v0[k], v1[k], v2[k]};
            Butterfly<3, DIR, V>::run(u);
            v0[k] = u[0];
            v1[k] = cmul(u[1], f1[T * k]);  // W_N^(j + T k)
            v2[k] = cmul(u[2], f2[T * k]);  // W_N^(2 (j + T k))
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            V* v = r == 0 ? v0 : (r == 1 ? v1 : v2);
            run_stages<V, PT, 0, DIR, CB, false, false, TW_LDS, false, 1, 1, NWV>(v, ldstw, lds, j, c);
#pragma unroll
            for (int k = 0; k < E; ++k) gstore<NTS>(op + (long long)ouni[k] + (long long)r * omap.stride, VT::to_g(cscale(v[k], sc)));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Column tiles whose INPUT side is the transposed one ([..][z][kx], kx fastest: the inverse X pass; reference fftX backward,
// fft_mpi_3d_api.cpp:553-570) -- the mirror image of the staged transposing store of fft_tiles_kernel (round 4).  On that side
// the columns of a tile are far apart and the FFT index is the contiguous one, so fft_tiles_kernel reads 128-byte pieces in fp64
// and cannot use column pairs at all in fp32 (two adjacent columns are not adjacent in memory: those launches ran on the scalar
// float2 kernel at half the rate).  Here a thread loads 16-byte memory elements in LINEAR order of the [scalar column][kx] tile
// (a wave reads 1 KiB runs), the workgroup stages them through an LDS image with one row per scalar column, and every thread
// picks its points j + T k of its column (pair) out of the image; the transform and the plain contiguous-column store are those of
// fft_tiles_kernel.  Single-block maps on both sides, whole tiles, one tile per workgroup at a time (fast path only); the output side
// may carry rotated rows per point (RotMap mode 2: the send buffer of a P > 1 backward plan).
// imap: stride 1 per MEMORY ELEMENT along kx, cstride per SCALAR column; itile: a_stride per slice, b_stride per column of V.
struct TuneTransposedLoad : TuneDefault {
    static constexpr bool OSTAGE = true;  // (sizes the LDS for the image: the same [CB * LANES][N + OPAD] rows as the staged store)
};
template <class V, class P, int CB, int DIR, bool ROT>
__global__ void __attribute__((amdgpu_flat_work_group_size(1, CB * P::T), amdgpu_waves_per_eu(1)))
fft_tload_tiles_kernel(const typename VecTraits<V>::G* in, typename VecTraits<V>::G* out, const typename VecTraits<V>::W* __restrict__ tw,
                       AxisMap imap, AxisMap omap, TileMap itile, TileMap otile, unsigned ntiles, unsigned tiles_per_a, unsigned a_first,
                       double scale, RotMap rm) {
    using KG = KernelGeom<V, P, CB, 1, TuneTransposedLoad>;
    using VT = VecTraits<V>;
    using W = typename VT::W;
    using GV = typename VT::G;
    constexpr int E = P::E, T = P::T, GT = CB * T, N = P::N, LANES = VT::LANES, ON = N / LANES, ROW = N + KG::OPAD;
    static_assert(KG::OSTAGE && KG::PH == 1 && !KG::PAD && !KG::WAVE_LOCAL, "transposed load: multi-stage plans on block-wide one-phase tiles");
    static_assert(ON % GT == 0 || GT % ON == 0, "transposed load: power-of-two geometry");
    extern __shared__ __attribute__((aligned(16))) char dfft_smem[];
    W* ldstw = reinterpret_cast<W*>(dfft_smem);
    V* lds = reinterpret_cast<V*>(dfft_smem + KG::TW_BYTES);
    const int tid = threadIdx.x, c = tid % CB, j = tile_j<CB, KG::NW>(tid);
    constexpr int TWN = KG::TWMODE == TW_REG ? KG::TWN : 0;
    W        twreg[TWN > 0 ? TWN : 1];
    const W* twr = twreg;
    if constexpr (KG::TWMODE == TW_GLOBAL) {
        twr = tw;
    } else if constexpr (KG::TWMODE == TW_LDS) {
#if DFFT_TW_STAGE_MAJOR
        fill_stage_major<W, P, 0, DIR, KG::NW>(ldstw, tw, tid, GT);
#else
        for (int i = tid; i < N; i += GT) {
            W w = tw[i];
            if (DIR < 0) w.y = -w.y;
            ldstw[i] = w;
        }
#endif
        __syncthreads();
        twr = ldstw;
    } else {
        load_twiddles<W, P, 0, DIR, TuneTransposedLoad::TWPOW>(twreg, tw, j);
    }
    constexpr bool TWPOW = TuneTransposedLoad::TWPOW && KG::TWMODE == TW_REG;
    // load step k: linear element tid + GT k of the tile = memory element (lin % ON) of scalar column lin / ON
    const long long ibase = ON % GT == 0 ? (long long)tid : (long long)(tid % ON) + (long long)(tid / ON) * imap.cstride;
    auto in_off = [&](int k) -> long long {
        const int oc = (GT * k) / ON;  // exact in both cases
        return ibase + (ON % GT == 0 ? (GT * k) % ON : 0) + (long long)oc * imap.cstride;
    };
    const int scol0 = ON % GT == 0 ? 0 : tid / ON, se0 = ON % GT == 0 ? tid : tid % ON;  // image position of load step 0
    const typename real_of<W>::type sc = (typename real_of<W>::type)scale;
    const unsigned or0 = (unsigned)(j * omap.stride + c * omap.cstride), ostep = (unsigned)(T * omap.stride);
    const int rot_j = ROT ? (rm.rot * j) & rm.mask : 0, rot_t = ROT ? (rm.rot * T) & rm.mask : 0;
    auto in_ptr = [&](unsigned t) -> const GV* {
        const unsigned al = t / tiles_per_a, b = t - al * tiles_per_a, a = al + a_first;
        return in + (long long)a * itile.a_stride + (long long)(b * CB) * itile.b_stride;
    };
    GV raw[E], rawn[E];
    unsigned t = blockIdx.x;
    if (t < ntiles) {
        const GV* ip = in_ptr(t);
#pragma unroll
        for (int k = 0; k < E; ++k) raw[k] = gload<true>(ip + in_off(k));
    }
    for (; t < ntiles; t += gridDim.x) {
        const unsigned tn = t + gridDim.x;
        if (tn < ntiles) {  // the next tile's loads complete underneath this tile's exchanges
            const GV* ip = in_ptr(tn);
#pragma unroll
            for (int k = 0; k < E; ++k) rawn[k] = gload<true>(ip + in_off(k));
        }
        W* img = reinterpret_cast<W*>(lds);
        __syncthreads();  // the previous tile's last exchange is no longer read
#pragma unroll
        for (int k = 0; k < E; ++k) {
            const int col = scol0 + (GT * k) / ON, e = se0 + (ON % GT == 0 ? (GT * k) % ON : 0);
            *reinterpret_cast<GV*>(img + col * ROW + e * LANES) = raw[k];
        }
        __syncthreads();
        V v[E];
#pragma unroll
        for (int k = 0; k < E; ++k) {
            if constexpr (LANES == 2) {
                const W a0 = img[(c * 2) * ROW + j + T * k], a1 = img[(c * 2 + 1) * ROW + j + T * k];
                v[k] = VT::from_g(GV{a0.x, a0.y, a1.x, a1.y});
            } else {
                v[k] = img[c * ROW + j + T * k];
            }
        }
        __syncthreads();  // the image is read before the first exchange overwrites it
        run_stages<V, P, 0, DIR, CB, false, false, KG::TWMODE, TWPOW, 1, 1, KG::NW>(v, twr, lds, j, c);
        const unsigned al = t / tiles_per_a, b = t - al * tiles_per_a, a = al + a_first;
        const int      ob0 = (int)(b * CB);
        GV*            op = out + (long long)a * otile.a_stride + (long long)ob0 * otile.b_stride;
        unsigned       o0 = or0;
        asm volatile("" : "+v"(o0));  // per tile, not a loop invariant (see fft_tiles_kernel)
        // the prefetched tile is waited for BEFORE this tile's stores are issued (it has had the exchanges to arrive), so that the
        // stores drain underneath the next tile instead of in front of it (one vmcnt for loads and stores)
#pragma unroll
        for (int k = 0; k < E; ++k) raw[k] = rawn[k];
#pragma unroll
        for (int k = 0; k < E; ++k) {
            long long off = (long long)(o0 + (unsigned)k * ostep);
            if constexpr (ROT) off += (long long)(((ob0 + rot_j + k * rot_t) & rm.mask) - ob0);
            gstore<false>(op + off, VT::to_g(cscale(v[k], sc)));
        }
    }
}

struct DeviceInfo {
    int cus;
};
inline const DeviceInfo& device_info() {
    static thread_local int cached_dev = -1;
    static thread_local DeviceInfo di;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev != cached_dev) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess) di.cus = p.multiProcessorCount;
        else di.cus = 256;
        cached_dev = dev;
    }
    return di;
}

inline hipError_t launch_debug(hipError_t e, const char* what, int lds, int threads) {
    if (getenv("DFFT_DEBUG"))
        fprintf(stderr, "[dfft] %s failed: %d (%s), lds=%d threads=%d\n", what, (int)e, hipGetErrorString(e), lds, threads);
    return e;
}

template <class V, class P, int CB, int G, int DIR, bool GENERAL, class Tune = TuneDefault>
hipError_t launch_variant(const FftLaunch& L, hipStream_t stream, int* blocks_per_cu_out = nullptr) {
    using KG = KernelGeom<V, P, CB, G, Tune>;
    auto kern = fft_tiles_kernel<V, P, CB, G, DIR, GENERAL, Tune>;
    // Per-device one-time set-up (function attributes are per device context): LDS opt-in and resident blocks per CU.
    // (several device threads may launch the same instantiation at once: published with release / read with acquire, set up
    // under a lock)
    static std::atomic<int> blocks_per_cu[64];
    static std::mutex       setup_mutex;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (blocks_per_cu[dev].load(std::memory_order_acquire) == 0) {
        std::lock_guard<std::mutex> lk(setup_mutex);
        if (blocks_per_cu[dev].load(std::memory_order_relaxed) == 0) {
        if (KG::LDS_BYTES > 64 * 1024) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)KG::LDS_BYTES);
            if (e != hipSuccess) return launch_debug(e, "hipFuncSetAttribute", (int)KG::LDS_BYTES, KG::THREADS);
        }
        int occ = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, KG::THREADS, KG::LDS_BYTES);
        if (e != hipSuccess) {
            // advisory only (the grid-stride loop is correct for any grid): fall back to the LDS / wave-slot bound
            (void)launch_debug(e, "hipOccupancyMaxActiveBlocksPerMultiprocessor", (int)KG::LDS_BYTES, KG::THREADS);
            (void)hipGetLastError();
            const int waves = (KG::THREADS + 63) / 64;
            occ = 32 / waves;
            if (KG::LDS_BYTES > 0 && (int)(160 * 1024 / KG::LDS_BYTES) < occ) occ = (int)(160 * 1024 / KG::LDS_BYTES);
        }
        blocks_per_cu[dev].store(occ > 0 ? occ : 1, std::memory_order_release);
        }
    }
    if (blocks_per_cu_out) *blocks_per_cu_out = blocks_per_cu[dev].load(std::memory_order_relaxed);
    const long long nblocks_needed = (L.ntiles + G - 1) / G;
    int bpc = blocks_per_cu[dev].load(std::memory_order_relaxed);
    if (L.blocks_per_cu_limit > 0 && L.blocks_per_cu_limit < bpc) bpc = L.blocks_per_cu_limit;
    long long grid = (long long)device_info().cus * bpc;
    if (L.grid_limit > 0 && grid > L.grid_limit) grid = L.grid_limit;
    if (grid > nblocks_needed) grid = nblocks_needed;
    if (grid < 1) return hipSuccess;
    (void)hipGetLastError();  // drop any stale error of this thread (other libraries share the runtime)
    using GV = typename VecTraits<V>::G;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(KG::THREADS), KG::LDS_BYTES, stream, (const GV*)L.in, (GV*)L.out,
                       (const typename VecTraits<V>::W*)L.tw, L.imap, L.omap, L.itile, L.otile, (unsigned)L.ntiles,
                       (unsigned)L.tiles_per_a, L.ncols, (unsigned)L.a_first, L.scale == 0.0 ? 1.0 : L.scale, L.rot);
    e = hipGetLastError();
    if (e != hipSuccess) return launch_debug(e, "kernel launch", (int)KG::LDS_BYTES, KG::THREADS);
    return hipSuccess;
}

template <class V, class P, int CB, int DIR, bool NT, bool ROT = false, int WPC = 1> hipError_t launch_dual(const FftLaunch& L, hipStream_t stream) {
    using VT = VecTraits<V>;
    using W = typename VT::W;
    using GV = typename VT::G;
    constexpr size_t LDS_BYTES = DualGeom<V, P, CB, WPC>::LDS_BYTES;
    auto kern = fft_dual_tiles_kernel<V, P, CB, DIR, NT, ROT, WPC>;
    static std::atomic<bool> attr_set[64];
    static std::mutex        setup_mutex;
    int         dev = 0;
    hipError_t  e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lk(setup_mutex);
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return launch_debug(e, "hipFuncSetAttribute", (int)LDS_BYTES, CB * P::T);
        attr_set[dev].store(true, std::memory_order_release);
    }
    const long long tiles_per_a = L.ncols / (2 * CB), ntiles = L.na * tiles_per_a;
    if (ntiles <= 0) return hipSuccess;
    if (ntiles >= (1ll << 31)) return hipErrorInvalidValue;
    long long grid = (long long)device_info().cus * WPC;
    if (L.grid_limit > 0 && grid > L.grid_limit) grid = L.grid_limit;
    if (grid > ntiles) grid = ntiles;
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(CB * P::T), LDS_BYTES, stream, (const GV*)L.in, (GV*)L.out, (const W*)L.tw, L.imap,
                       L.omap, L.itile, L.otile, (unsigned)ntiles, (unsigned)tiles_per_a, (unsigned)L.a_first, L.scale == 0.0 ? 1.0 : L.scale, L.rot);
    e = hipGetLastError();
    if (e != hipSuccess) return launch_debug(e, "kernel launch", (int)LDS_BYTES, CB * P::T);
    return hipSuccess;
}

template <class V, class PH, int CB, int DIR, bool NTL, bool NTS, bool BIN, bool BOUT, int ROT = 0, bool TOUT = false> hipError_t launch_dif2(const FftLaunch& L, hipStream_t stream) {
    using VT = VecTraits<V>;
    using W = typename VT::W;
    using GV = typename VT::G;
    constexpr size_t LDS_BYTES = Dif2Geom<V, PH, CB, TOUT>::LDS_BYTES;
    auto kern = fft_dif2_tiles_kernel<V, PH, CB, DIR, NTL, NTS, BIN, BOUT, ROT, TOUT>;
    static std::atomic<int> blocks_per_cu[64];  // 0 = not set up on that device yet
    static std::mutex       setup_mutex;
    int         dev = 0;
    hipError_t  e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (blocks_per_cu[dev].load(std::memory_order_acquire) == 0) {
        std::lock_guard<std::mutex> lk(setup_mutex);
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return launch_debug(e, "hipFuncSetAttribute", (int)LDS_BYTES, CB * PH::T);
        int occ = 0;  // a half tile of 64 KiB (1024 points) leaves room for a second workgroup when the registers allow it
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, CB * PH::T, LDS_BYTES) != hipSuccess) {
            (void)hipGetLastError();
            occ = 1;
        }
        blocks_per_cu[dev].store(occ > 0 ? occ : 1, std::memory_order_release);
    }
    const long long tiles_per_a = L.ncols / CB, ntiles = L.na * tiles_per_a;
    if (ntiles <= 0) return hipSuccess;
    if (ntiles >= (1ll << 31)) return hipErrorInvalidValue;
    long long grid = (long long)device_info().cus * blocks_per_cu[dev].load(std::memory_order_relaxed);
    if (L.grid_limit > 0 && grid > L.grid_limit) grid = L.grid_limit;
    if (grid > ntiles) grid = ntiles;
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(CB * PH::T), LDS_BYTES, stream, (const GV*)L.in, (GV*)L.out, (const W*)L.tw, L.imap,
                       L.omap, L.itile, L.otile, (unsigned)ntiles, (unsigned)tiles_per_a, (unsigned)L.a_first, L.scale == 0.0 ? 1.0 : L.scale, L.rot);
    e = hipGetLastError();
    if (e != hipSuccess) return launch_debug(e, "kernel launch", (int)LDS_BYTES, CB * PH::T);
    return hipSuccess;
}

template <class V, class PT, int CB, int DIR, bool NTL, bool NTS, int ROT = 0> hipError_t launch_dif3(const FftLaunch& L, hipStream_t stream) {
    using VT = VecTraits<V>;
    using W = typename VT::W;
    using GV = typename VT::G;
    constexpr size_t LDS_BYTES = Dif3Geom<V, PT, CB>::LDS_BYTES;
    auto kern = fft_dif3_tiles_kernel<V, PT, CB, DIR, NTL, NTS, ROT>;
    static std::atomic<int> blocks_per_cu[64];  // 0 = not set up on that device yet
    static std::mutex       setup_mutex;
    int         dev = 0;
    hipError_t  e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (blocks_per_cu[dev].load(std::memory_order_acquire) == 0) {
        std::lock_guard<std::mutex> lk(setup_mutex);
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return launch_debug(e, "hipFuncSetAttribute", (int)LDS_BYTES, CB * PT::T);
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, CB * PT::T, LDS_BYTES) != hipSuccess) {
            (void)hipGetLastError();
            occ = 1;
        }
        blocks_per_cu[dev].store(occ > 0 ? occ : 1, std::memory_order_release);
    }
    const long long tiles_per_a = L.ncols / CB, ntiles = L.na * tiles_per_a;
    if (ntiles <= 0) return hipSuccess;
    if (ntiles >= (1ll << 31)) return hipErrorInvalidValue;
    long long grid = (long long)device_info().cus * blocks_per_cu[dev].load(std::memory_order_relaxed);
    if (L.grid_limit > 0 && grid > L.grid_limit) grid = L.grid_limit;
    if (grid > ntiles) grid = ntiles;
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(CB * PT::T), LDS_BYTES, stream, (const GV*)L.in, (GV*)L.out, (const W*)L.tw, L.imap,
                       L.omap, L.itile, L.otile, (unsigned)ntiles, (unsigned)tiles_per_a, (unsigned)L.a_first, L.scale == 0.0 ? 1.0 : L.scale, L.rot);
    e = hipGetLastError();
    if (e != hipSuccess) return launch_debug(e, "kernel launch", (int)LDS_BYTES, CB * PT::T);
    return hipSuccess;
}

template <class V, class P, int CB, int DIR, bool ROT> hipError_t launch_tload(const FftLaunch& L, hipStream_t stream) {
    using VT = VecTraits<V>;
    using W = typename VT::W;
    using GV = typename VT::G;
    using KG = KernelGeom<V, P, CB, 1, TuneTransposedLoad>;
    constexpr size_t LDS_BYTES = KG::LDS_BYTES;
    auto kern = fft_tload_tiles_kernel<V, P, CB, DIR, ROT>;
    static std::atomic<int> blocks_per_cu[64];  // 0 = not set up on that device yet
    static std::mutex       setup_mutex;
    int         dev = 0;
    hipError_t  e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (blocks_per_cu[dev].load(std::memory_order_acquire) == 0) {
        std::lock_guard<std::mutex> lk(setup_mutex);
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return launch_debug(e, "hipFuncSetAttribute", (int)LDS_BYTES, CB * P::T);
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, CB * P::T, LDS_BYTES) != hipSuccess) {
            (void)hipGetLastError();
            occ = 1;
        }
        blocks_per_cu[dev].store(occ > 0 ? occ : 1, std::memory_order_release);
    }
    const long long tiles_per_a = L.ncols / CB, ntiles = L.na * tiles_per_a;
    if (ntiles <= 0) return hipSuccess;
    if (ntiles >= (1ll << 31)) return hipErrorInvalidValue;
    long long grid = (long long)device_info().cus * blocks_per_cu[dev].load(std::memory_order_relaxed);
    if (L.grid_limit > 0 && grid > L.grid_limit) grid = L.grid_limit;
    if (grid > ntiles) grid = ntiles;
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(CB * P::T), LDS_BYTES, stream, (const GV*)L.in, (GV*)L.out, (const W*)L.tw, L.imap,
                       L.omap, L.itile, L.otile, (unsigned)ntiles, (unsigned)tiles_per_a, (unsigned)L.a_first, L.scale == 0.0 ? 1.0 : L.scale, L.rot);
    e = hipGetLastError();
    if (e != hipSuccess) return launch_debug(e, "kernel launch", (int)LDS_BYTES, CB * P::T);
    return hipSuccess;
}

template <int X> struct ConstMax1 { static constexpr int value = X < 1 ? 1 : X; };

// Columns per tile for the column kernel: a full 128-byte line per row segment (8 fp64 / 16 fp32 complex).  Tiles up to
// 128 KiB exchange through the LDS in one go (blocks up to 1024 threads); tiles up to 256 KiB in two phases
// (KernelGeom::PH) when the block stays at 512 threads and a thread's points fit 96 VGPRs, so that the kernel keeps the
// 256-register budget without spilling (1280 and 1536 points: 2.9 -> 3.2, 3.2 -> 3.6 TB/s; 2048 points with 32 points per
// thread spills 190 registers and drops from 3.1 to 2.3 TB/s, so it keeps half-line tiles).  Otherwise halve.
template <class V, class P> constexpr int cols_per_tile() {
    int cb = 128 / (int)sizeof(V);
    constexpr bool two_phase_ok = P::E % 2 == 0 && P::E * (int)sizeof(V) / 4 <= 96;
    while (cb > 1) {
        const long long bytes = (long long)P::N * cb * (long long)sizeof(V);
        const bool      fits = bytes <= 128 * 1024 ? cb * P::T <= 1024 : (two_phase_ok && bytes <= 256 * 1024 && cb * P::T <= 512);
        if (fits) break;
        cb /= 2;
    }
    return cb;
}

// staging the transposed store pays when the tile rows are full 128-byte lines and the image fits the LDS
// (N = 1024 fp64: 131 KiB, measured 3.4 -> 4.4 TB/s together with 16 points per thread; round 1)
template <class V, class P, int CBO = 0> constexpr bool can_stage_store() {
    constexpr int CBC = CBO > 0 ? CBO : cols_per_tile<V, P>();
    constexpr int GC = ConstMax1<256 / (CBC * P::T)>::value;
    constexpr int L = VecTraits<V>::LANES;
    // (column pairs also stage with half-line tiles: their alternative is the scalar float2 kernel, 2048-point X pass)
    constexpr int PH = (size_t)P::N * CBC * sizeof(V) > 128 * 1024 ? 2 : 1;  // KernelGeom::PH
    // (... and so do the half-line tiles of the lean lengths, CBO > 0: what the image buys is 1 KiB runs on the OUTPUT side)
    return P::S > 1 && CBC * (int)sizeof(V) >= ((L == 2 || CBO > 0) ? 64 : 128) && P::N % L == 0 &&
           (size_t)(P::N + (L == 2 ? 2 : 1)) * CBC / PH * GC * sizeof(V) <= 144 * 1024;
}

// the staged transposed LOAD (fft_tload_tiles_kernel) exists where the staged store's image fits, on one-phase tiles of at least 256
// threads whose linear element order splits evenly over the workgroup.  -DDFFT_TLOAD=0 compiles the path out (A/B builds).
#ifndef DFFT_TLOAD
#define DFFT_TLOAD 1
#endif
template <class V, class P> constexpr bool can_tload() {
    constexpr int CBC = cols_per_tile<V, P>();
    constexpr int GT = CBC * P::T, ON = P::N / VecTraits<V>::LANES;
    // fp32 column pairs always (their alternative is the scalar kernel); fp64 only up to 256 points -- measured (round 4,
    // profiles/r04/experiments/lib_ab_transposed_load.log, backward plans, inverse X pass): fp32 1024^3 7.38 -> 4.02 ms, config 5's rank
    // at P = 8 3.65 -> 2.62, 512^3 fp32 0.52 -> 0.455; fp64 256^3 0.1045 -> 0.0893 but 512^3 0.84 -> 0.99 and 1024 x 768 x 512 2.49 ->
    // 2.76 (fp64 already reads whole lines through fft_tiles_kernel and pays here for the extra LDS round trip).
    constexpr bool type_ok = VecTraits<V>::LANES == 2 || P::N <= 256;
    return DFFT_TLOAD && type_ok && sizeof(V) == 16 && P::S > 1 && can_stage_store<V, P>() && GT >= 256 && GT <= 1024 && (ON % GT == 0 || GT % ON == 0) &&
           (size_t)P::N * CBC * sizeof(V) <= 128 * 1024;
}
// run-time side of the same rule (fast path only: single-block maps, whole tiles, rotation -- if any -- per point on the output side)
template <class V, class P> bool tload_applies(const FftLaunch& L) {
    if constexpr (!can_tload<V, P>()) {
        return false;
    } else {
        constexpr int CBC = cols_per_tile<V, P>();
        const bool    general = (L.ncols % CBC) != 0 || L.imap.last_delta != 0 || L.omap.last_delta != 0;
        const bool    tin = L.imap.nblk == 1 && L.imap.sub <= 1 && L.imap.stride == 1 && L.imap.cstride != 1;
        const bool    cout_ = L.omap.nblk == 1 && L.omap.sub <= 1 && L.omap.cstride == 1 && L.otile.b_stride == 1;
        const bool    rot = L.rot.in_mode != 0 || L.rot.out_mode != 0;
        const bool    rot_ok = !rot || (L.rot.in_mode == 0 && L.rot.out_mode == 2);
        return !general && tin && cout_ && rot_ok && axis_max_offset(L.omap, P::N) < (1ll << 32);
    }
}

// Column launches describe the work as `na` slices of `ncols` columns; the tile geometry follows from the variant's CB.
// Row launches (contiguous FFTs, one per tile) may use a plan of their own: the row kernel keeps no column tile in LDS, so
// it can afford more points per thread than the column kernel of the same length (2048: 32 points, one wave per FFT).
// (Row FFTs that need a whole multi-wave workgroup -- 4096, 2401, 3125 points -- were also tried with the register budget
// bounded so that two workgroups are resident per CU: every such build spills (100-430 B) and runs at 0.5-0.95 of the
// unbounded kernels' rate -- 4096: 8979 vs 9196 GFlop/s, 2401: 4324 vs 7095, 3125: 3242 vs 7602; profiles/r03/experiments/
// rows_ab_*.log -- so they keep the default: one fat workgroup per CU, no scratch.)
template <class V, class P> hipError_t launch_rows(const FftLaunch& L, hipStream_t stream) {
    static_assert(VecTraits<V>::LANES == 1, "column pairs exist only for the column kernel");
    constexpr int GR = ConstMax1<256 / P::T>::value;  // ~256 threads per block
    if (L.ntiles <= 0) return hipSuccess;
    if (L.ntiles >= (1ll << 31)) return hipErrorInvalidValue;
    if (L.hints & FFT_HINT_STREAM_IN) {
        if (L.dir > 0) return launch_variant<V, P, 1, GR, +1, false, TuneStreamIn>(L, stream);
        return launch_variant<V, P, 1, GR, -1, false, TuneStreamIn>(L, stream);
    }
    if (L.dir > 0) return launch_variant<V, P, 1, GR, +1, false>(L, stream);
    return launch_variant<V, P, 1, GR, -1, false>(L, stream);
}

// PH: plan of the half length (N / 2 points) for lengths whose non-transposing column passes run DIF-split (void: none)
// CBO: columns per tile when not cols_per_tile() -- the half-line tiles of the "lean" lengths, see below.
#ifndef DFFT_LEAN_COLS
#define DFFT_LEAN_COLS 1
#endif
template <class V, class P, class PH = void, int CBO = 0> hipError_t launch_plan(const FftLaunch& Lin, hipStream_t stream) {
    constexpr int CBC = CBO > 0 ? CBO : cols_per_tile<V, P>();
    constexpr int GC = ConstMax1<256 / (CBC * P::T)>::value;  // column kernel
    if (!Lin.cols) return hipErrorInvalidValue;  // rows: launch_rows
    FftLaunch L = Lin;
    L.tiles_per_a = (L.ncols + CBC - 1) / CBC;
    L.ntiles = L.na * L.tiles_per_a;
    if (L.ntiles <= 0) return hipSuccess;
    if (L.ntiles >= (1ll << 31)) return hipErrorInvalidValue;
    const bool general = (L.ncols % CBC) != 0 || L.imap.last_delta != 0 || L.omap.last_delta != 0;
    // rotated rows (RotMap): whole tiles move inside their rows, so the rotation and the row length must be multiples of the
    // widest tile used below (a full line), sides must have unit column stride, and the launch must be a fast-path one
    const bool rot = L.rot.in_mode != 0 || L.rot.out_mode != 0;
    if (rot) {
        constexpr int LINE = 128 / (int)sizeof(V);
        const bool ok = !general && L.rot.rot > 0 && L.rot.rot % LINE == 0 && (L.rot.mask + 1) % LINE == 0 && ((L.rot.mask + 1) & L.rot.mask) == 0 &&
                        L.ncols == L.rot.mask + 1 && (L.rot.in_mode == 0 || (L.imap.cstride == 1 && L.itile.b_stride == 1)) &&
                        (L.rot.out_mode == 0 || (L.omap.cstride == 1 && L.otile.b_stride == 1));
        if (!ok) return hipErrorInvalidValue;
    }
    // "Lean" lengths (round 6): 1000, 1280 and 1536 points hold 10-24 points of 16 bytes per thread on full-line tiles of 512-800
    // threads, and every variant that needs per-point offsets (packed / two-level maps, rotated rows, uneven slabs, the staged store)
    // kept 52-320 bytes in scratch (profiles/r06/kernel_resources.txt of library b1622e22).  Full-line tiles stay where they are
    // clean -- the Plain twins on single-block maps, the staged store of 1536 fp64 points -- and everything else runs on HALF-line
    // tiles (4 columns: 256-400 threads, twice the register budget, no scratch; the XCD-aware tile order of fft_tiles_kernel gives
    // the two tiles of a line to one L2).  -DDFFT_LEAN_COLS=0 restores the full-line variants (A/B builds).
    // -DDFFT_LEAN_EXTRA=1 (experiment, round 6): the same half-line tiles for the packed / rotated variants of 768 and 1024 points, which
    // do NOT spill -- does a 256-thread workgroup with two or three of its kind resident per CU beat the 512-thread full-line tile?
    // Measured and NOT adopted (profiles/r06/experiments/lib_ab_lean_extra_768_1024.log, bit-identical): config 4's rank at P = 8
    // t0 0.56 -> 0.60 ms, t3 0.35 -> 0.46; 1024^3 fp32 at P = 4 1.45 / 0.82 -> 1.58 / 1.16.  What the lean lengths gain is the scratch
    // they lose, not the geometry.
#ifndef DFFT_LEAN_EXTRA
#define DFFT_LEAN_EXTRA 0
#endif
    constexpr bool lean_extra = DFFT_LEAN_EXTRA && CBO == 0 && sizeof(V) == 16 && (P::N == 768 || P::N == 1024) && CBC >= 2;
    if constexpr (lean_extra) {
        const bool single = L.imap.nblk == 1 && L.imap.sub <= 1 && L.omap.nblk == 1 && L.omap.sub <= 1;
        if (!general && (rot || !single)) return launch_plan<V, P, void, CBC / 2>(Lin, stream);
    }
    constexpr bool lean_len = DFFT_LEAN_COLS && CBO == 0 && sizeof(V) == 16 && (P::N == 1000 || P::N == 1280 || P::N == 1536) && CBC >= 2;
    if constexpr (lean_len) {
        const bool single = L.imap.nblk == 1 && L.imap.sub <= 1 && L.omap.nblk == 1 && L.omap.sub <= 1;
        const bool staged = can_stage_store<V, P>() && !general && L.omap.nblk == 1 && L.omap.stride == 1 && L.omap.cstride != 1;
        if (!general && !rot && single && !staged) {
            if (L.dir > 0) {
                if (L.hints & FFT_HINT_STREAM_OUT) return launch_variant<V, P, CBC, GC, +1, false, Plain<TuneColsStreamOut>>(L, stream);
                return launch_variant<V, P, CBC, GC, +1, false, Plain<TuneCols>>(L, stream);
            }
            if (L.hints & FFT_HINT_STREAM_IN) return launch_variant<V, P, CBC, GC, -1, false, Plain<TuneColsStreamIn>>(L, stream);
            return launch_variant<V, P, CBC, GC, -1, false, Plain<TuneCols>>(L, stream);
        }
        if constexpr (P::N == 1536 && VecTraits<V>::LANES == 1 && can_stage_store<V, P>()) {
            if (staged && !rot) {
                if (L.dir > 0) return launch_variant<V, P, CBC, GC, +1, false, TuneTransposedStore>(L, stream);
                return launch_variant<V, P, CBC, GC, -1, false, TuneTransposedStore>(L, stream);
            }
        }
        return launch_plan<V, P, PH, CBC / 2>(Lin, stream);
    } else if constexpr (CBC * P::T <= 1024) {
        // transposed INPUT side (the inverse X pass): staged load
        if constexpr (CBO == 0 && can_tload<V, P>()) {
            if (tload_applies<V, P>(L)) {
                if (rot) return L.dir > 0 ? launch_tload<V, P, CBC, +1, true>(L, stream) : launch_tload<V, P, CBC, -1, true>(L, stream);
                return L.dir > 0 ? launch_tload<V, P, CBC, +1, false>(L, stream) : launch_tload<V, P, CBC, -1, false>(L, stream);
            }
        }
        if constexpr (VecTraits<V>::LANES == 2) {  // column pairs reach a transposed input side only through that kernel (make_pair_launch)
            if (L.imap.nblk == 1 && L.imap.stride == 1 && L.imap.cstride != 1 && !(L.omap.nblk == 1 && L.omap.stride == 1 && L.omap.cstride != 1))
                return hipErrorInvalidValue;
        }
        // transposing store (unit stride along the FFT index on the output side, columns far apart): staged variant
        using TT = TuneTransposedStore;
        using TTF = TuneTransposedStoreFull;
        constexpr bool can_stage = can_stage_store<V, P, CBO>();
        const bool staged = can_stage && !general && L.omap.nblk == 1 && L.omap.stride == 1 && L.omap.cstride != 1;
        // 16 points of 16 bytes per thread on full-line tiles (the 1024-point forward X pass, fp64 and fp32 pairs): whole-tile prefetch
        // when the input side is a single-block map (TuneTransposedStoreFull); FFT_HINT_HALF_PREFETCH / FFT_HINT_EARLY_WAIT (the plan's
        // DFFT_X_VARIANT=half|full|fullearly|early) select the other variants for A/B measurements.
        constexpr bool can_full = can_stage && P::E == 16 && sizeof(V) == 16 && CBC * (int)sizeof(V) == 128 && CBC * P::T * GC <= 512;
        using TTFE = TuneTransposedStoreFullEarly;
        using TTE = TuneTransposedStoreEarly;
        const bool single_in = L.imap.nblk == 1 && L.imap.sub <= 1 && L.imap.last_delta == 0;
        const bool want_early = (L.hints & FFT_HINT_EARLY_WAIT) != 0;
        const int  full = (can_full && staged && L.dir > 0 && single_in && !(L.hints & FFT_HINT_HALF_PREFETCH)) ? (want_early ? 2 : 1) : 0;
        // (8 points per thread, the 512-point X pass: the early wait is a measurement switch only)
        constexpr bool can_early = can_stage && P::N == 512 && sizeof(V) == 16;
        const bool early = can_early && staged && L.dir > 0 && want_early;
        // half-line tiles (2048 points) with the transposing store: pair the two tiles of every 128-byte line in one workgroup
        // when columns are adjacent in memory on the input side (forward X pass)
        constexpr bool can_dual = P::S > 1 && sizeof(V) == 16 && 2 * CBC * sizeof(V) == 128 && (P::N & (P::N - 1)) == 0 && (P::N / VecTraits<V>::LANES) % (CBC * P::T) == 0 &&
                                  (size_t)P::N * CBC * sizeof(V) + (size_t)P::N * sizeof(typename VecTraits<V>::W) <= 160 * 1024;
        // 1024 points (round 4, -DDFFT_DUAL_1024=1): the full-line tile (128 KiB + table) leaves one 512-thread workgroup per CU; paired
        // half-line tiles of 4 columns need 64 KiB + table = 80 KiB and 256 threads, so TWO workgroups are resident per CU and one
        // transforms while the other one's loads and stores are in flight (each still fetches whole 128-byte lines: it owns both
        // tiles of a line).  Measured and NOT adopted (900 parity cases green; profiles/r04/experiments/lib_ab_dual_1024_two_per_cu.log):
        // slower everywhere -- config 4's shape X pass 2.72 -> 3.27 ms, fp32 1.19 -> 1.60, its rank at P = 8 0.373 -> 0.392: two resident
        // workgroups that each have nothing in flight underneath their transforms lose to one workgroup with a whole-tile prefetch.
#ifndef DFFT_DUAL_1024
#define DFFT_DUAL_1024 0
#endif
        constexpr bool can_dual2 = DFFT_DUAL_1024 && P::N == 1024 && P::E == 16 && sizeof(V) == 16 && (P::N / VecTraits<V>::LANES) % (4 * P::T) == 0;
        if constexpr (can_dual2) {
            constexpr int CBH = 4;
            const bool    transposed = L.omap.nblk == 1 && L.omap.stride == 1 && L.omap.cstride != 1;
            if (!general && transposed && L.imap.blk % P::T == 0 && L.ncols % (2 * CBH) == 0 && L.imap.cstride == 1 && L.itile.b_stride == 1 &&
                (!rot || (L.rot.in_mode == 2 && L.rot.out_mode == 0))) {
                if (rot) return L.dir > 0 ? launch_dual<V, P, CBH, +1, true, true, 2>(L, stream) : launch_dual<V, P, CBH, -1, true, true, 2>(L, stream);
                return L.dir > 0 ? launch_dual<V, P, CBH, +1, true, false, 2>(L, stream) : launch_dual<V, P, CBH, -1, true, false, 2>(L, stream);
            }
        }
        // The same kernel for the 1024-point forward X pass (round 6, config 4's t3): two 512-point half transforms through a 64 KiB tile
        // + 16 KiB table = 80 KiB, so TWO workgroups share a CU and one transforms while the other's loads and stores are in flight (at most
        // 128 registers each: Dif2Geom::TWO_PER_CU) -- instead of TuneTransposedStoreFull's one 128 KiB tile with a whole-tile prefetch.
        // Measured (profiles/r06/experiments/lib_ab_x_pass_1024_dif2.log, three processes each, DFFT_X_DIF2=0 against the default): fp64
        // config 4's rank at P = 8 0.342 -> 0.306 ms, at P = 4 0.682 -> 0.582, one GPU 2.56 -> 2.32, 1024 x 1024 x 512 at P = 8 0.453 -> 0.393;
        // fp32 pairs with rotated rows (P > 1) 0.175 -> 0.162 and 0.828 -> 0.795, but on the padded hand-over buffer of a single-GPU plan
        // 1.08 -> 1.19 and 2.81 -> 3.13 -- and a rule "pairs only with rotated rows" would give up the bit-identity of rotated and
        // un-rotated pipelines (test_rotated_exchange_rows_vs_oracle), so 1024-point pairs stay on TuneTransposedStoreFull.
        // -DDFFT_X_DIF2_1024=0 compiles it out.
#ifndef DFFT_X_DIF2_1024
#define DFFT_X_DIF2_1024 1
#endif
#ifndef DFFT_DIF2_HALF
#define DFFT_DIF2_HALF 0
#endif
        if constexpr (!std::is_void<PH>::value && ((VecTraits<V>::LANES == 2 && P::N >= 2048) || (DFFT_X_DIF2_1024 && P::N == 1024 && VecTraits<V>::LANES == 1))) {  // (fp64 2048: 108-116 bytes of scratch next to the rotated-row image -- stays on the paired tiles)
            // forward X pass of lengths whose full-line tile does not fit the LDS (2048 points): DIF-split full-line tiles with the
            // staged transposed store (round 6; fft_dif2_tiles_kernel, TOUT).  DFFT_X_DIF2=0: the paired half-line tiles of rounds 2-5.
            static const bool x_dif2 = [] {
                const char* e = getenv("DFFT_X_DIF2");
                return !(e && *e == '0');
            }();
            constexpr int CBF = 128 / (int)sizeof(V);
            const bool    transposed = L.omap.nblk == 1 && L.omap.stride == 1 && L.omap.cstride != 1 && L.omap.last_delta == 0;
            constexpr bool want = P::N >= 2048 || VecTraits<V>::LANES == 1;
            if (x_dif2 && want && L.dir > 0 && !general && transposed && L.imap.cstride == 1 && L.itile.b_stride == 1 && L.imap.nblk == 1 && L.imap.sub <= 1 && L.ncols % CBF == 0 &&
                L.imap.last_delta == 0 && (axis_max_offset(L.imap, P::N / 2) + L.ncols) * (long long)sizeof(V) < (1ll << 32) && L.itile.b_stride == 1 &&
                (!rot || (L.rot.in_mode == 2 && L.rot.out_mode == 0))) {
                // (the slab side of an X pass is a single-block map: offsets kk * step, no table)
                if (rot) return launch_dif2<V, PH, CBF, +1, true, true, false, false, 2, true>(L, stream);
                return launch_dif2<V, PH, CBF, +1, true, true, false, false, 0, true>(L, stream);
            }
        }
        if constexpr (can_dual) {
            static const bool no_dual = [] {  // DFFT_NO_DUAL=1: A/B switch for measurements
                const char* e = getenv("DFFT_NO_DUAL");
                return e && *e && *e != '0';
            }();
            const bool transposed = L.omap.nblk == 1 && L.omap.stride == 1 && L.omap.cstride != 1;
            if (!no_dual && !general && transposed && L.imap.blk % P::T == 0 && L.ncols % (2 * CBC) == 0 && L.imap.cstride == 1 && L.itile.b_stride == 1 &&
                (!rot || (L.rot.in_mode == 2 && L.rot.out_mode == 0))) {
                if (rot) return L.dir > 0 ? launch_dual<V, P, CBC, +1, true, true>(L, stream) : launch_dual<V, P, CBC, -1, true, true>(L, stream);
                return L.dir > 0 ? launch_dual<V, P, CBC, +1, true>(L, stream) : launch_dual<V, P, CBC, -1, true>(L, stream);
            }
        }
#ifndef DFFT_DIF3
#define DFFT_DIF3 0
#endif
#if DFFT_DIF3
        if constexpr (P::N == 768 && VecTraits<V>::LANES == 1) {
            // radix-3 split tiles (fft_dif3_tiles_kernel): three 256-point sub-transforms through a 32 KiB tile, three workgroups per CU
            using PT3 = Plan<256, 8, 8, 8, 4>;
            constexpr int CB3 = 128 / (int)sizeof(V);
            static const bool dif3 = [] {
                const char* e = getenv("DFFT_DIF3");
                return e && *e == '1';
            }();
            const bool lines = L.imap.cstride == 1 && L.itile.b_stride == 1 && L.omap.cstride == 1 && L.otile.b_stride == 1;
            const bool even3 = L.ncols % CB3 == 0 && L.imap.last_delta == 0 && L.omap.last_delta == 0;
            const bool fit3 = axis_max_offset(L.imap, P::N) < (1ll << 32) && axis_max_offset(L.omap, P::N) < (1ll << 32);
            const bool rot3 = !rot || (L.rot.in_mode != 2 && L.rot.out_mode != 2);
            if (dif3 && !general && lines && even3 && fit3 && rot3 && L.imap.blk % PT3::T == 0 && L.omap.blk % (3 * PT3::T) == 0) {
                const bool s_in = (L.hints & FFT_HINT_STREAM_IN) != 0, s_out = (L.hints & FFT_HINT_STREAM_OUT) != 0;
                if (rot) {
                    if (L.dir > 0) return s_out ? launch_dif3<V, PT3, CB3, +1, false, true, 1>(L, stream) : launch_dif3<V, PT3, CB3, +1, false, false, 1>(L, stream);
                    return s_in ? launch_dif3<V, PT3, CB3, -1, true, false, 1>(L, stream) : launch_dif3<V, PT3, CB3, -1, false, false, 1>(L, stream);
                }
                if (L.dir > 0) return s_out ? launch_dif3<V, PT3, CB3, +1, false, true>(L, stream) : launch_dif3<V, PT3, CB3, +1, false, false>(L, stream);
                return s_in ? launch_dif3<V, PT3, CB3, -1, true, false>(L, stream) : launch_dif3<V, PT3, CB3, -1, false, false>(L, stream);
            }
        }
#endif
        if constexpr (!std::is_void<PH>::value) {
            // full-line tiles through the DIF split whenever both sides keep the 8 (16 fp32) columns of a line together
            constexpr int CBF = 128 / (int)sizeof(V);
            static_assert(2 * PH::N == P::N && CBF * PH::T <= 1024, "half plan of the wrong length");
            static const bool no_dif2 = [] {  // DFFT_NO_DIF2=1: A/B switch for measurements
                const char* e = getenv("DFFT_NO_DIF2");
                return e && *e && *e != '0';
            }();
            // both sides keep the 8 (16 fp32) columns of a line together.  (The transposed [..][z][kx] side of the forward X pass
            // was tried on this kernel too -- a thread's results 2m and 2m + 1 are neighbours in kx, stored 16 bytes at a time by the
            // two half transforms: 2.2 TB/s against 4.6-4.8 for the staged / paired-tile kernels at 1024 and 2048 points,
            // profiles/r03/experiments/xpass_variants.log -- and removed.)
            const bool lines_in = L.imap.cstride == 1 && L.itile.b_stride == 1;
            const bool lines_out = L.omap.cstride == 1 && L.otile.b_stride == 1;
            constexpr bool transposed_out = false;
            const bool even = L.ncols % CBF == 0 && L.imap.last_delta == 0 && L.omap.last_delta == 0;
            const bool sin = (L.hints & FFT_HINT_STREAM_IN) != 0, sout = (L.hints & FFT_HINT_STREAM_OUT) != 0;
            // (the kernel keeps its per-point block offsets in 32 bits)
            const bool fits32 = axis_max_offset(L.imap, P::N) < (1ll << 32) && axis_max_offset(L.omap, P::N) < (1ll << 32);
            const bool rot_tile = rot && L.rot.in_mode != 2 && L.rot.out_mode != 2;  // whole tiles move (Y passes)
            // DFFT_DIF2_MIN=<n>: lengths from n on use the split.  Default 1024 since round 4 (two 512-point halves through a 64 KiB
            // tile): t0 -2 ... -3 % wherever the Y axis is 1024 points long (1024^3 fp32 6.11 -> 5.93 ms, 512 x 1024 x 512 fp64 2.97 ->
            // 2.90), backward plans unchanged within the noise; 916 parity cases (profiles/r04/experiments/dif2_1024_point_*.log).
            // DFFT_DIF2_MIN=2048 restores the 16-point-per-thread full-line kernel for 1024 points.
            static const int dif2_min = [] {
                const char* e = getenv("DFFT_DIF2_MIN");
                return e ? atoi(e) : 1024;
            }();
            if (!no_dif2 && P::N >= dif2_min && lines_in && (lines_out || transposed_out) && even && fits32 && L.imap.blk % PH::T == 0 &&
                L.omap.blk % (2 * PH::T) == 0 && (!rot || rot_tile)) {
                if (rot) {
                    if (L.dir > 0) {
#if DFFT_DIF2_HALF
                        // experiment (round 6): the packing Y pass of 2048-point column pairs on HALF-line tiles (4 pairs: 64 KiB half tile +
                        // 16 KiB table, two workgroups per CU) -- DFFT_Y_DIF2_HALF=1 selects it
                        if constexpr (VecTraits<V>::LANES == 2 && P::N == 2048) {
                            static const bool half = [] {
                                const char* e = getenv("DFFT_Y_DIF2_HALF");
                                return e && *e == '1';
                            }();
                            if (half && sout && L.ncols % (CBF / 2) == 0) return launch_dif2<V, PH, CBF / 2, +1, false, true, true, true, 1>(L, stream);
                        }
#endif
                        if (sout) return launch_dif2<V, PH, CBF, +1, false, true, true, true, 1>(L, stream);
                        return launch_dif2<V, PH, CBF, +1, false, false, true, true, 1>(L, stream);
                    }
                    if (sin) return launch_dif2<V, PH, CBF, -1, true, false, true, true, 1>(L, stream);
                    return launch_dif2<V, PH, CBF, -1, false, false, true, true, 1>(L, stream);
                }
                if (L.dir > 0) {
                    if (sout) return launch_dif2<V, PH, CBF, +1, false, true, true, true>(L, stream);
                    if (sin) return launch_dif2<V, PH, CBF, +1, true, false, true, true>(L, stream);
                    return launch_dif2<V, PH, CBF, +1, false, false, true, true>(L, stream);
                }
                if (sin) return launch_dif2<V, PH, CBF, -1, true, false, true, true>(L, stream);
                if (sout) return launch_dif2<V, PH, CBF, -1, false, true, true, true>(L, stream);
                return launch_dif2<V, PH, CBF, -1, false, false, true, true>(L, stream);
            }
        }
        if (rot) {  // the rotated twins of the fast-path variants: forward Y stores rotated tiles (0, 1), forward X loads rotated
                    // points (2, 0); backward X stores rotated points (0, 2), backward Y loads rotated tiles (1, 0)
            const int im = L.rot.in_mode, om = L.rot.out_mode;
            if (L.dir > 0 && im == 0 && om == 1) {
                if (L.hints & FFT_HINT_STREAM_OUT) return launch_variant<V, P, CBC, GC, +1, false, WithRot<TuneColsStreamOut, 0, 1>>(L, stream);
                return launch_variant<V, P, CBC, GC, +1, false, WithRot<TuneCols, 0, 1>>(L, stream);
            }
            if (L.dir > 0 && im == 2 && om == 0) {
                if constexpr (can_full) {
                    if (full == 2) return launch_variant<V, P, CBC, GC, +1, false, WithRot<TTFE, 2, 0>>(L, stream);
                    if (full == 1) return launch_variant<V, P, CBC, GC, +1, false, WithRot<TTF, 2, 0>>(L, stream);
                }
                if constexpr (can_stage)
                    if (staged) return launch_variant<V, P, CBC, GC, +1, false, WithRot<TT, 2, 0>>(L, stream);
                return launch_variant<V, P, CBC, GC, +1, false, WithRot<TuneCols, 2, 0>>(L, stream);
            }
            if (L.dir < 0 && im == 0 && om == 2) {
                if (L.hints & FFT_HINT_STREAM_IN) return launch_variant<V, P, CBC, GC, -1, false, WithRot<TuneColsStreamIn, 0, 2>>(L, stream);
                return launch_variant<V, P, CBC, GC, -1, false, WithRot<TuneCols, 0, 2>>(L, stream);
            }
            if (L.dir < 0 && im == 1 && om == 0) {
                if (L.hints & FFT_HINT_STREAM_IN) return launch_variant<V, P, CBC, GC, -1, false, WithRot<TuneColsStreamIn, 1, 0>>(L, stream);
                return launch_variant<V, P, CBC, GC, -1, false, WithRot<TuneCols, 1, 0>>(L, stream);
            }
            return hipErrorInvalidValue;
        }
        // single-block maps on both sides: the PLAIN twins of the column variants, for the lengths that need the registers
        constexpr bool can_plain = CBO == 0 && (P::N == 1000 || P::N == 1280 || P::N == 1536);
        if constexpr (can_plain) {
            const bool single = L.imap.nblk == 1 && L.imap.sub <= 1 && L.omap.nblk == 1 && L.omap.sub <= 1;
            const bool is_staged = can_stage && staged;
            if (!general && single && !is_staged) {
                if (L.dir > 0) {
                    if (L.hints & FFT_HINT_STREAM_OUT) return launch_variant<V, P, CBC, GC, +1, false, Plain<TuneColsStreamOut>>(L, stream);
                    return launch_variant<V, P, CBC, GC, +1, false, Plain<TuneCols>>(L, stream);
                }
                if (L.hints & FFT_HINT_STREAM_IN) return launch_variant<V, P, CBC, GC, -1, false, Plain<TuneColsStreamIn>>(L, stream);
                return launch_variant<V, P, CBC, GC, -1, false, Plain<TuneCols>>(L, stream);
            }
        }
        // a directly accessed transposed side (see TransposedSide): the twin without the wave-interleaved labelling, where the two differ
        const bool tr_side = !(L.imap.cstride == 1 && L.itile.b_stride == 1) || (!(L.omap.cstride == 1 && L.otile.b_stride == 1) && !(can_stage && staged));
        using TSCols = std::conditional_t<(KernelGeom<V, P, CBC, GC, TuneCols>::NW > 0), TransposedSide<TuneCols>, TuneCols>;
        using TSColsIn = std::conditional_t<(KernelGeom<V, P, CBC, GC, TuneColsStreamIn>::NW > 0), TransposedSide<TuneColsStreamIn>, TuneColsStreamIn>;
        using TSColsOut = std::conditional_t<(KernelGeom<V, P, CBC, GC, TuneColsStreamOut>::NW > 0), TransposedSide<TuneColsStreamOut>, TuneColsStreamOut>;
        if (tr_side) {
            if (L.dir > 0) {
                if (general) return launch_variant<V, P, CBC, GC, +1, true, TSCols>(L, stream);
                if (L.hints & FFT_HINT_STREAM_OUT) return launch_variant<V, P, CBC, GC, +1, false, TSColsOut>(L, stream);
                return launch_variant<V, P, CBC, GC, +1, false, TSCols>(L, stream);
            }
            if (general) return launch_variant<V, P, CBC, GC, -1, true, TSCols>(L, stream);
            if (L.hints & FFT_HINT_STREAM_IN) return launch_variant<V, P, CBC, GC, -1, false, TSColsIn>(L, stream);
            return launch_variant<V, P, CBC, GC, -1, false, TSCols>(L, stream);
        }
        if (L.dir > 0) {
            if (general) return launch_variant<V, P, CBC, GC, +1, true, TuneCols>(L, stream);
            if constexpr (can_full) {
                if (full == 2) return launch_variant<V, P, CBC, GC, +1, false, TTFE>(L, stream);
                if (full == 1) return launch_variant<V, P, CBC, GC, +1, false, TTF>(L, stream);
            }
            if constexpr (can_early)
                if (early) return launch_variant<V, P, CBC, GC, +1, false, TTE>(L, stream);
            if constexpr (can_stage)
                if (staged) return launch_variant<V, P, CBC, GC, +1, false, TT>(L, stream);
            if (L.hints & FFT_HINT_STREAM_OUT) return launch_variant<V, P, CBC, GC, +1, false, TuneColsStreamOut>(L, stream);
            return launch_variant<V, P, CBC, GC, +1, false, TuneCols>(L, stream);
        }
        if (general) return launch_variant<V, P, CBC, GC, -1, true, TuneCols>(L, stream);
        if constexpr (can_stage)
            if (staged) return launch_variant<V, P, CBC, GC, -1, false, TT>(L, stream);
        if (L.hints & FFT_HINT_STREAM_IN) return launch_variant<V, P, CBC, GC, -1, false, TuneColsStreamIn>(L, stream);
        return launch_variant<V, P, CBC, GC, -1, false, TuneCols>(L, stream);
    } else {
        return hipErrorInvalidValue;
    }
}

// fp32 column launches whose two sides both keep adjacent columns adjacent in memory (or store them transposed through
// the staged image) run on column PAIRS: 16 bytes per lane like the fp64 kernels, packed-fp32 butterflies.  Rewrites the
// launch into units of pairs; returns false when the launch is not eligible (odd counts/strides, unaligned bases, a
// transposed side that cannot be staged) and the scalar float2 kernel has to take it.
template <class P> bool make_pair_launch(const FftLaunch& L, FftLaunch& out) {
    static const bool disabled = [] {  // DFFT_NO_PAIRS=1: A/B switch for measurements
        const char* e = getenv("DFFT_NO_PAIRS");
        return e && *e && *e != '0';
    }();
    if (disabled || !L.cols || L.dtype != F32 || L.ncols < 2 || (L.ncols & 1)) return false;
    if (((uintptr_t)L.in | (uintptr_t)L.out) & 15) return false;
    constexpr int CBC = cols_per_tile<cpair, P>();
    auto even = [](long long v) { return (v & 1) == 0; };
    out = L;
    out.ncols = L.ncols / 2;
    if (L.rot.in_mode != 0 || L.rot.out_mode != 0) {  // rotated rows: the same rotation in units of pairs
        if ((L.rot.rot & 1) || !(L.rot.mask & 1)) return false;
        out.rot.rot = L.rot.rot / 2;
        out.rot.mask = (L.rot.mask + 1) / 2 - 1;
    }
    // contiguous-column side: every stride halves, columns stay unit-stride
    auto contiguous = [&](const AxisMap& m, const TileMap& t, AxisMap& mo, TileMap& to) {
        if (m.cstride != 1 || t.b_stride != 1) return false;
        if (!even(m.stride) || !even(m.blk_stride) || !even(m.last_delta) || !even(t.a_stride) || !even(m.sub_stride)) return false;
        mo = m;
        mo.stride = m.stride / 2;
        mo.blk_stride = m.blk_stride / 2;
        mo.sub_stride = m.sub_stride / 2;
        mo.last_delta = m.last_delta / 2;
        to.a_stride = t.a_stride / 2;
        to.b_stride = 1;
        return true;
    };
    if (!contiguous(L.imap, L.itile, out.imap, out.itile)) {
        // transposed input side (unit stride along the FFT index, columns far apart: the inverse X pass): only through the staged
        // load, whose map is (memory element along kx, scalar column) -- fft_tload_tiles_kernel
        if constexpr (can_tload<cpair, P>()) {
            if (!contiguous(L.omap, L.otile, out.omap, out.otile)) return false;
            if (L.imap.nblk != 1 || L.imap.sub > 1 || L.imap.stride != 1 || L.imap.cstride == 1 || L.imap.last_delta != 0 || (L.imap.blk & 1)) return false;
            if (!even(L.imap.cstride) || !even(L.itile.a_stride) || !even(L.itile.b_stride)) return false;
            out.imap = L.imap;
            out.imap.blk = L.imap.blk / 2;           // memory elements along the FFT index
            out.imap.stride = 1;
            out.imap.cstride = L.imap.cstride / 2;   // per scalar column
            out.itile.a_stride = L.itile.a_stride / 2;
            out.itile.b_stride = L.itile.b_stride;   // per PAIR of columns = 2 * (b_stride / 2)
            return tload_applies<cpair, P>(out);
        }
        return false;
    }
    if (contiguous(L.omap, L.otile, out.omap, out.otile)) return true;
    // transposed output side (unit stride along the FFT index, columns far apart): only through the staged store, whose
    // map is (memory element along kx, scalar column) -- see fft_tiles_kernel
    if constexpr (can_stage_store<cpair, P>()) {
        const bool general = (out.ncols % CBC) != 0 || L.imap.last_delta != 0;
        if (general || L.omap.nblk != 1 || L.omap.stride != 1 || L.omap.last_delta != 0) return false;
        if (!even(L.omap.cstride) || !even(L.otile.a_stride) || !even(L.otile.b_stride)) return false;
        out.omap = L.omap;
        out.omap.blk = L.omap.blk / 2;           // memory elements along the FFT index
        out.omap.stride = 1;
        out.omap.cstride = L.omap.cstride / 2;   // per scalar column
        out.otile.a_stride = L.otile.a_stride / 2;
        out.otile.b_stride = L.otile.b_stride;   // per PAIR of columns = 2 * (b_stride / 2)
        return true;
    }
    return false;
}

// Scalar float2 columns: the fall-back of fp32 column launches that make_pair_launch turns down (an odd Z length gives odd column
// counts and strides; a caller's buffer that is only 8-byte aligned).  Up to round 5 they went through launch_plan like the other two
// element types -- 16-19 variants per length on 16-column tiles, i.e. 1024-thread workgroups with a 128-register budget for every
// length from 512 points on, and 150 of those kernels kept 12-204 bytes in scratch (profiles/r05/kernel_resources.txt).  Round 6
// gives the fall-back a geometry of its own instead: at most 512 threads per workgroup (the 256-register budget), no register
// prefetch, tid / CB butterfly ids, direct stores on a transposed side -- and eight kernels per length: whole-tile and ragged
// (GENERAL) in both directions plus the four rotated-row twins a P > 1 plan can ask for.  No scratch in any of them
// (profiles/r06/kernel_resources.txt); the stream hints are ignored (they select cache policies, not results).
struct TuneScalar : TuneDefault {
    static constexpr bool NO_OWNED = true;
};
template <class P> constexpr int cols_per_tile_scalar32() {
    int cb = 128 / (int)sizeof(float2);
    // (20 and more points per thread: 256 threads, i.e. one wave per SIMD and its 512 registers -- the unrolled exchanges of the five-
    // and six-stage plans keep their LDS addresses in registers)
    constexpr int MAXT = P::E >= 20 ? 256 : 512;
    while (cb > 1 && (cb * P::T > MAXT || (long long)P::N * cb * (long long)sizeof(float2) > 128 * 1024)) cb /= 2;
    return cb;
}
template <class P> hipError_t launch_scalar32(const FftLaunch& Lin, hipStream_t stream) {
    using V = float2;
    constexpr int CBC = cols_per_tile_scalar32<P>();
    constexpr int GC = ConstMax1<256 / (CBC * P::T)>::value;
    static_assert(CBC * P::T * GC <= 512, "the scalar fall-back keeps the 256-register budget");
    if (!Lin.cols) return hipErrorInvalidValue;
    FftLaunch L = Lin;
    L.tiles_per_a = (L.ncols + CBC - 1) / CBC;
    L.ntiles = L.na * L.tiles_per_a;
    if (L.ntiles <= 0) return hipSuccess;
    if (L.ntiles >= (1ll << 31)) return hipErrorInvalidValue;
    const bool general = (L.ncols % CBC) != 0 || L.imap.last_delta != 0 || L.omap.last_delta != 0;
    if (L.rot.in_mode != 0 || L.rot.out_mode != 0) {  // same admission rule as launch_plan (whole 128-byte lines move)
        constexpr int LINE = 128 / (int)sizeof(V);
        const bool ok = !general && L.rot.rot > 0 && L.rot.rot % LINE == 0 && (L.rot.mask + 1) % LINE == 0 && ((L.rot.mask + 1) & L.rot.mask) == 0 &&
                        L.ncols == L.rot.mask + 1 && (L.rot.in_mode == 0 || (L.imap.cstride == 1 && L.itile.b_stride == 1)) &&
                        (L.rot.out_mode == 0 || (L.omap.cstride == 1 && L.otile.b_stride == 1));
        if (!ok) return hipErrorInvalidValue;
        const int im = L.rot.in_mode, om = L.rot.out_mode;
        if (L.dir > 0 && im == 0 && om == 1) return launch_variant<V, P, CBC, GC, +1, false, WithRot<TuneScalar, 0, 1>>(L, stream);
        if (L.dir > 0 && im == 2 && om == 0) return launch_variant<V, P, CBC, GC, +1, false, WithRot<TuneScalar, 2, 0>>(L, stream);
        if (L.dir < 0 && im == 0 && om == 2) return launch_variant<V, P, CBC, GC, -1, false, WithRot<TuneScalar, 0, 2>>(L, stream);
        if (L.dir < 0 && im == 1 && om == 0) return launch_variant<V, P, CBC, GC, -1, false, WithRot<TuneScalar, 1, 0>>(L, stream);
        return hipErrorInvalidValue;
    }
    if (L.dir > 0) {
        if (general) return launch_variant<V, P, CBC, GC, +1, true, TuneScalar>(L, stream);
        return launch_variant<V, P, CBC, GC, +1, false, TuneScalar>(L, stream);
    }
    if (general) return launch_variant<V, P, CBC, GC, -1, true, TuneScalar>(L, stream);
    return launch_variant<V, P, CBC, GC, -1, false, TuneScalar>(L, stream);
}

// One entry point per FFT length, explicitly instantiated in dfft_fft_inst.hip (split over translation units so the
// gfx950 code objects build in parallel).
template <int N> hipError_t launch_n(const FftLaunch& L, hipStream_t stream);

}  // namespace dfft
