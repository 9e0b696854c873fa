// dfft_zy.hip -- t0 (batched 2D YZ FFT of every owned plane; reference fftZY, fft_mpi_3d_api.cpp:466-522) as ONE persistent launch.
//
// The plan's default t0 alternates two launches per 256 MiB Infinity-Cache chunk of planes (Z rows, then Y columns: 16 launches at
// 512^3 fp64).  This kernel runs the same phases -- all row units of a chunk, then all column units of the same chunk -- inside
// one launch: work items are handed out by a global ticket counter in that order; a row unit publishes its rows with write-through
// (sc1) stores and one relaxed agent-scope increment of its plane's counter; a column unit starts when its plane's counter shows
// that every row unit of the plane has published, and reads the rows with sc1 loads (MI355X_MICROARCH.md, inter-workgroup
// hand-off: "16 B sc1 stores AND sc1 loads" -- no fences; a release / acquire fence per unit cost 0.15-0.2 ms of a 1.37 ms stage
// in the experiments of round 2, profiles/r02/experiments/zy_stream_first_run.log).  What is saved is the 15 launch boundaries
// inside the stage (the phases overlap at their edges instead of draining the GPU): 1.365 -> 1.337 ms in the experiment that
// preceded this file.  The inverse transform runs the mirror image (column units first, then row units into the result buffer).
//
// The arithmetic is the library's own (run_stages of dfft_fft_impl.h with the same plans and register twiddles): the eager-publish
// kernel is bit-identical to the two-launch path, the lazy-publish one (the default, see LAZY below) agrees
// with it to the last bit or two and is deterministic -- tests/test_gpu_parity.py::test_one_launch_t0_is_bit_identical.
//
// Deadlock freedom: tickets are taken in order by RUNNING workgroups only; a producer unit never waits; a consumer unit waits only
// for producer units with smaller tickets, and a workgroup never holds an unprocessed item -- nor, in the lazy form, an unpublished
// producer unit -- while it waits (the next item is prefetched only if its dependency is already satisfied).  Every spin is bounded
// by a number of POLLS (spin_polls, about 1-3 us each: seconds in all -- not by the wall clock, which keeps running while the
// process's queues are preempted by another process on the GPU, a profiler or a debugger); a time-out sets ctl->error and the
// host-visible error word, every workgroup leaves, and every later launch on the same control block refuses to run (the error is
// sticky, and a launch whose first ticket is implausible for its ticket_base -- counters out of step with the host -- raises it
// too), so a desynchronised stage can never return garbage silently.  The host sees the pinned word without a copy: it reports
// the failure at the next execute / synchronisation, re-runs a host-synchronised execute on the two-launch path where the input
// is still intact, and never uses the stage again (dfft_plan.cpp, zy_check).
//
// Scope: fp64, Y and Z lengths of 256 or 512 points (one wavefront per row FFT, 8 points per thread on both axes); single-GPU fused
// plans (hand-over buffer or bufferDev1 as w) and P > 1 fused plans with even splits (the Y side then reads / writes the packed,
// row-rotated exchange layout), whole slabs or the X-plane parts of the overlapped pipeline.  Everything else keeps the chunk loop.
// The plan uses the stage by itself on every single-GPU plan it is built for (since round 4: two-line column tiles for 256-point Y
// axes, ZyTile below) and on P > 1 plans with 512 x 512-point planes (dfft_plan.cpp, profiles/r04/README.md sections 1-2).
#include "dfft_fft_impl.h"
#include "dfft_plans.h"
#include "dfft_zy.h"

namespace dfft {

namespace {

typedef unsigned zy_u32x4 __attribute__((ext_vector_type(4)));
// cache policy of the send-buffer stores of SIG launches: sc1 (16, write-through) | nt (2, streaming: the send buffer is not read again by
// this device and must not push the Z -> Y intermediate out of the Infinity Cache).  -DDFFT_ZY_SIG_AUX=16: sc1 alone (A/B)
#ifndef DFFT_ZY_SIG_AUX
#define DFFT_ZY_SIG_AUX 18
#endif
#define DFFT_ZY_AGENT __HIP_MEMORY_SCOPE_AGENT

// DIR = +1: producers = Z rows (src -> w), consumers = Y columns (w in place)
// DIR = -1: producers = Y columns (w in place), consumers = Z rows (w -> dst)
// PACK: the column side that is not w is the packed exchange layout of a P > 1 plan (forward: Y columns w -> packed send buffer,
// backward: packed receive buffer -> w) described by the launcher's axis map `pk`, with the rows rotated per plane (RotMap mode 1)
// LAZY (the plan's default for packed and un-packed launches, DFFT_ZY_LAZY=0 selects the eager form): gfx9 counts loads and stores in one vmcnt, so "wait for the prefetched unit" after a
// unit's stores have been issued means "drain those stores" -- and a producer unit drains them again before it publishes.  The lazy
// form waits for everything that is in flight (the previous unit's stores, the next unit's loads) after a unit's ARITHMETIC, when
// it has had a whole unit's exchanges to complete, publishes the PREVIOUS producer unit there, and only then issues this unit's
// stores, which drain underneath the next unit.  A workgroup never enters a blocking wait with an unpublished unit (flush first).
// Column tiles: one cache line (8 columns) for 512-point Y axes -- 512 threads, 64 KiB units.  A 256-point Y axis would make that a
// 256-thread workgroup with 32 KiB units, whose per-unit overhead (ticket, dependency poll, quiet point) weighs twice as much: its
// un-packed launches take tiles of TWO lines (16 columns), which restores the 512 threads and the 64 KiB.  Packed launches keep one
// line (a rotated tile position is a multiple of one line, so a wider tile could straddle the end of a row).
template <class PY, bool PACK> struct ZyTile {
    static constexpr bool WIDE = !PACK && PY::T < 64 && (size_t)PY::N * (8 * (64 / PY::T)) * sizeof(double2) <= 128 * 1024;  // (768 points: 192 KiB, no)
    static constexpr int  CB = WIDE ? 8 * (64 / PY::T) : 8;
    static constexpr int THREADS = CB * PY::T;
};

// SIGN: direction of the TRANSFORM (sign of the twiddles); DIR above is the STRUCTURE -- which kind of unit produces.  They differ in one
// case (round 5): the inverse YZ stage of a single-GPU plan runs rows first (DIR = +1, SIGN = -1).  The two 1-D transforms of a plane
// commute, and columns-first makes the column units READ the hand-over buffer from HBM in 128-byte pieces one row apart (the inverse X pass
// has just written all of it; the counters show 40 % more read latency behind the L2 than in the forward stage, profiles/r05/README.md
// section 4) where rows-first reads whole rows from HBM and leaves the strided side to the cache-resident chunk, like the forward stage.
// SIG (round 6; forward packed lazy launches of the overlapped pipeline): the launch covers the WHOLE slab and tells the exchange stream
// when an X-plane part of the send buffer is complete -- every column unit, once its stores have left the CU (they are write-through
// sc1 buffer stores here, so "left the CU" means "in memory", like the Z -> Y hand-over), adds one to its part's counter
// part_done[(plane - plane0) / part_planes]; a one-wave kernel on the exchange stream (zy_part_wait_kernel) waits for the count of
// the part's column units and the part's exchange follows it in stream order.  One launch, the phases of the serial plan, no
// per-part launch boundaries.
template <class PZ, class PY, int DIR, bool PACK, bool LAZY = false, int SIGN = DIR, bool SIG = false>
__global__ void __attribute__((amdgpu_flat_work_group_size(ZyTile<PY, PACK>::THREADS, ZyTile<PY, PACK>::THREADS), amdgpu_waves_per_eu(1)))
zy_chunk_kernel(const double2* src, double2* w, double2* dst, ZyCtl* ctl, const double2* __restrict__ twz, const double2* __restrict__ twy,
                long long src_plane, long long w_plane, long long dst_plane, unsigned plane0, unsigned nplanes, unsigned chunk,
                unsigned ticket_base, unsigned done_base, AxisMap pk, long long pk_plane, RotMap rm, unsigned* err_host, unsigned spin_polls, unsigned need,
                unsigned* part_done, unsigned part_planes
#if DFFT_ZY_ROW_PITCH
                , unsigned wpitch
#endif
                ) {
    using V = double2;
    constexpr int CB = ZyTile<PY, PACK>::CB;  // columns per tile: one cache line, or two for 256-point Y axes (ZyTile)
    constexpr int THREADS = ZyTile<PY, PACK>::THREADS;
    // points per thread: 8 on both axes for 256 / 512 points; a 768-point Y axis (round 5, BASELINE config 4) holds 24 per thread in a
    // 256-thread workgroup -- one register set sized for the larger kind serves both kinds of unit
    constexpr int N2 = PZ::N, N1 = PY::N, EZ = PZ::E, EY = PY::E, E = EZ > EY ? EZ : EY, TZ = PZ::T, TY = PY::T;
    static_assert(TZ <= 64 && 64 % TZ == 0 && THREADS % TZ == 0, "rows: one FFT inside one wavefront");
    static_assert(!SIG || (PACK && LAZY && DIR > 0 && SIGN > 0), "part signalling: forward packed lazy launches");
    constexpr int GR = THREADS / TZ;  // rows per row unit
    static_assert(N1 % GR == 0 && N2 % CB == 0, "a plane must split into whole units");
    constexpr unsigned UZ = N1 / GR, UY = N2 / CB;
    constexpr unsigned UA = DIR > 0 ? UZ : UY, UB = DIR > 0 ? UY : UZ, BB = UA + UB;  // producer / consumer units per plane
    // twiddles exactly as the library's two-launch kernels of the same plans keep them (KernelGeom::TWMODE), so that the arithmetic is
    // the same to the bit: per-thread power sets in registers where a thread needs at most 16 of them, otherwise (768 points: 37) a
    // stage-major copy of the table in LDS, read in plain (non-power) form
    constexpr int      TWMZ = TwTotal<PZ, true>::value <= 16 ? TW_REG : TW_LDS, TWMY = TwTotal<PY, true>::value <= 16 ? TW_REG : TW_LDS;
    constexpr bool     TWPZ = TWMZ == TW_REG, TWPY = TWMY == TW_REG;
    constexpr size_t   TWZ_BYTES = TWMZ == TW_LDS ? (size_t)N2 * sizeof(V) : 0, TWY_BYTES = TWMY == TW_LDS ? (size_t)N1 * sizeof(V) : 0;
    constexpr int      ROW_LDS = N2 + N2 / 8;  // padded row (lds_index<1, true>)
// row pitch of w: the compile-time N2, or the launch parameter of the -DDFFT_ZY_ROW_PITCH=1 build (dfft_zy.h)
#if DFFT_ZY_ROW_PITCH
#define DFFT_ZY_WP ((int)wpitch)
#else
#define DFFT_ZY_WP N2
#endif

    extern __shared__ __attribute__((aligned(16))) char dfft_smem[];
    unsigned* shw = reinterpret_cast<unsigned*>(dfft_smem);  // [0] ticket broadcast, [1] dependency state
    V*        ldstwz = reinterpret_cast<V*>(dfft_smem + 64);
    V*        ldstwy = reinterpret_cast<V*>(dfft_smem + 64 + TWZ_BYTES);
    V*        lds = reinterpret_cast<V*>(dfft_smem + 64 + TWZ_BYTES + TWY_BYTES);

    const int tid = threadIdx.x;
    const int gz = tid / TZ, jz = tid % TZ;  // row unit: row gz of the unit, butterfly id jz
    // wave-interleaved butterfly ids of the column units (dfft_fft_impl.h, tile_j); not in the eager 768-point kernels -- the tests' bit-identity
    // reference, not a shipped path -- which have no register left for the swizzled exchange's second base
    constexpr int NWY = (!LAZY && PY::N == 768) ? 0 : owned_waves<V, CB, TY>();
    const int cy = tid % CB, jy = tile_j<CB, NWY>(tid);  // column unit: column cy of the tile, butterfly id jy
    V*        lds_row = lds + gz * ROW_LDS;

    constexpr int TWNZ = TWPZ ? TwTotal<PZ, true>::value : 0, TWNY = TWPY ? TwTotal<PY, true>::value : 0;
    V        twzr[TWNZ > 0 ? TWNZ : 1], twyr[TWNY > 0 ? TWNY : 1];
    const V *twzp = twzr, *twyp = twyr;
    if constexpr (TWPZ) load_twiddles<V, PZ, 0, SIGN, true>(twzr, twz, jz);
    else {
        fill_stage_major<V, PZ, 0, SIGN>(ldstwz, twz, tid, THREADS);
        twzp = ldstwz;
    }
    if constexpr (TWPY) load_twiddles<V, PY, 0, SIGN, true>(twyr, twy, jy);
    else {
        fill_stage_major<V, PY, 0, SIGN, NWY>(ldstwy, twy, tid, THREADS);
        twyp = ldstwy;
    }
    if constexpr (!TWPZ || !TWPY) __syncthreads();

    const unsigned CH = chunk;
    const unsigned nchunks = (nplanes + CH - 1) / CH;
    const unsigned total = nchunks * CH * BB;  // tickets (some of the last chunk's are empty); the host's ticket_base steps by this
    enum { NONE = 0, PROD = 1, CONS = 2 };
    struct Item {
        unsigned ticket, kind, plane, unit;
    };
    auto decode = [&](unsigned t) -> Item {
        Item it{t, NONE, 0u, 0u};
        if (t >= total) return it;
        const unsigned c = t / (CH * BB), r = t - c * (CH * BB);
        if (r < CH * UA) {
            const unsigned pl = c * CH + r / UA;
            if (pl < nplanes) it = Item{t, PROD, plane0 + pl, r % UA};
        } else {
            const unsigned r2 = r - CH * UA, pl = c * CH + r2 / UB;
            if (pl < nplanes) it = Item{t, CONS, plane0 + pl, r2 % UB};
        }
        return it;
    };
    auto share = [&](unsigned value_of_thread0) -> unsigned {
        if (tid == 0) shw[0] = value_of_thread0;
        __syncthreads();
        unsigned t = shw[0];
        if constexpr (LAZY) t = __builtin_amdgcn_readfirstlane(t);  // block-uniform: keep the decoded item in scalar registers
        __syncthreads();
        return t;
    };
    auto take = [&]() -> unsigned {  // thread 0: the next ticket (the atomic's latency hides behind the caller's work)
        // (the counters are never reset: launch g of a plan starts at ticket_base = g * total and done_base = g * UA, all in
        // wrapping 32-bit arithmetic, so no memset launch sits between two transforms)
        return tid == 0 ? __hip_atomic_fetch_add(&ctl->ticket, 1u, __ATOMIC_RELAXED, DFFT_ZY_AGENT) - ticket_base : 0u;
    };
    // dependency of a consumer unit: every producer unit of its plane has published -- `need` of them (= UA; the host's fault-injection
    // hook passes UA + 1, which no plane ever reaches: tests/test_gpu_parity.py::test_one_launch_t0_failure_is_loud_and_recovered).
    // wait = false: one poll only.
    // give up: sticky error in the control block (seen by every workgroup of this and of later launches) + the host-visible word
    auto raise = [&](unsigned code) {
        __hip_atomic_store(&ctl->error, code, __ATOMIC_RELAXED, DFFT_ZY_AGENT);
        __hip_atomic_store(err_host, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    auto ready = [&](const Item& it, bool wait) -> bool {
        if (it.kind != CONS) return true;
        if (tid == 0) {
            unsigned ok = __hip_atomic_load(&ctl->done[it.plane], __ATOMIC_RELAXED, DFFT_ZY_AGENT) - done_base >= need ? 1u : 0u;
            if (!ok && wait) {
                // bounded by polls this wave actually makes, not by elapsed time: a preempted process does not time out
                for (unsigned polls = 0;; ++polls) {
                    __builtin_amdgcn_s_sleep(1);
                    if (__hip_atomic_load(&ctl->done[it.plane], __ATOMIC_RELAXED, DFFT_ZY_AGENT) - done_base >= need) {
                        ok = 1u;
                        break;
                    }
                    if (__hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, DFFT_ZY_AGENT) != 0u) break;
                    if (polls >= spin_polls) {
                        raise(ZY_ERR_TIMEOUT);
                        break;
                    }
                }
            }
            shw[1] = ok;
        }
        __syncthreads();
        unsigned okw = shw[1];
        if constexpr (LAZY) okw = __builtin_amdgcn_readfirstlane(okw);
        const bool ok = okw != 0u;
        __syncthreads();
        return ok;
    };
    auto wrsrc = [&](unsigned plane) {  // buffer descriptor of one plane of w (offsets inside a plane fit 32 bits)
        return __builtin_amdgcn_make_buffer_rsrc((void*)(w + (long long)plane * w_plane), 0, (int)((size_t)N1 * DFFT_ZY_WP * sizeof(V)), 0x00020000);
    };
    // packed side (PACK): point jy + TY k of a column lies in block (TY k) / pk.blk of the map -- the launcher guarantees
    // pk.blk % TY == 0, so the block term is wave-uniform per k (computed once) -- plus one per-thread term and the tile's base
    // 64 threads per column (512 and 768 points): destination blocks are multiples of 32 rows (config 4: 96 rows at P = 8; the 32-row Y
    // sub-blocks of the overlapped pipeline of 512^3 at P = 8), not of the 64 a column's threads span, so the lower and the upper half of
    // the threads of a column each get a uniform term of their own (the same numbers when a block is whole 64-row strides)
    // (the eager 512-point kernels -- the tests' bit-identity reference -- have no registers for the second table: whole 64-row strides there)
    constexpr bool PK2 = PACK && TY == 64 && (LAZY || PY::N == 768);
    unsigned       pk_uni[PACK ? EY : 1], pk_uni1[PK2 ? EY : 1];
    if constexpr (PACK) {
#pragma unroll
        for (int k = 0; k < EY; ++k) {
            const int ib = (TY * k) / pk.blk;
            pk_uni[k] = (unsigned)(block_term(pk, ib) + (long long)(TY * k - ib * pk.blk) * pk.stride);
            if constexpr (PK2) {
                const int i1 = (TY * k + 32) / pk.blk;
                pk_uni1[k] = (unsigned)(block_term(pk, i1) + (long long)(TY * k + 32 - i1 * pk.blk) * pk.stride);
            }
        }
    }
    const bool      pk_hi = PK2 && jy >= 32;
    const long long pk_thr = PACK ? (long long)(PK2 ? (jy & 31) : jy) * pk.stride + cy : 0ll;
    auto            pk_off = [&](int k) -> unsigned {
        if constexpr (PK2) return pk_hi ? pk_uni1[k] : pk_uni[k];
        else return pk_uni[k];
    };
    auto            pk_base = [&](unsigned plane, unsigned un) -> long long {  // tile base: the plane's rows, rotated tile position
        int col = (int)(un * CB);
        if (rm.rot != 0) col = (col + rm.rot * (int)(plane + (unsigned)rm.a0)) & rm.mask;
        return (long long)plane * pk_plane + col + pk_thr;
    };
    constexpr bool ROWS_PRODUCE = DIR > 0;  // which kind of unit hands its results over through w
    // ---- row units (Z): forward src -> w (sc1 stores), backward w (sc1 loads) -> dst
    auto load_rows = [&](unsigned plane, unsigned un, V* d) {
        if constexpr (ROWS_PRODUCE) {
            const V* ip = src + (long long)plane * src_plane + (long long)(un * GR + gz) * N2 + jz;
#pragma unroll
            for (int k = 0; k < EZ; ++k) d[k] = gload<true>(ip + TZ * k);  // streamed input
        } else {
            const __amdgpu_buffer_rsrc_t rs = wrsrc(plane);
#pragma unroll
            for (int k = 0; k < EZ; ++k) {
                const unsigned elem = (unsigned)((un * GR + gz) * DFFT_ZY_WP + jz + TZ * k);
                d[k] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(elem * 16u), 0, 16 /* sc1 */));
            }
        }
    };
    auto store_rows = [&](unsigned plane, unsigned un, const V* v) {
        if constexpr (ROWS_PRODUCE) {
            const __amdgpu_buffer_rsrc_t rs = wrsrc(plane);
#pragma unroll
            for (int k = 0; k < EZ; ++k) {
                const unsigned elem = (unsigned)((un * GR + gz) * DFFT_ZY_WP + jz + TZ * k);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(zy_u32x4, v[k]), rs, (int)(elem * 16u), 0, 16 /* sc1 */);
            }
        } else {
            V* op = dst + (long long)plane * dst_plane + (long long)(un * GR + gz) * N2 + jz;
#pragma unroll
            for (int k = 0; k < EZ; ++k) op[TZ * k] = v[k];
        }
    };
    // ---- column units (Y), in place on w: forward sc1 loads / plain stores, backward plain loads / sc1 stores
    auto load_cols = [&](unsigned plane, unsigned un, V* d) {
        if constexpr (ROWS_PRODUCE) {
            const __amdgpu_buffer_rsrc_t rs = wrsrc(plane);
#pragma unroll
            for (int k = 0; k < EY; ++k) {
                const unsigned elem = (unsigned)((jy + TY * k) * DFFT_ZY_WP + un * CB + cy);
                d[k] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(elem * 16u), 0, 16 /* sc1 */));
            }
        } else if constexpr (PACK) {
            const V* ip = src + pk_base(plane, un);  // the receive buffer, written before this launch: streamed, plain visibility
#pragma unroll
            for (int k = 0; k < EY; ++k) d[k] = gload<true>(ip + pk_off(k));
        } else {
            const V* ip = w + (long long)plane * w_plane + (long long)jy * DFFT_ZY_WP + un * CB + cy;
#pragma unroll
            for (int k = 0; k < EY; ++k) d[k] = ip[(long long)(TY * k) * DFFT_ZY_WP];
        }
    };
    auto store_cols = [&](unsigned plane, unsigned un, const V* v) {
        if constexpr (ROWS_PRODUCE && PACK && SIG) {
            // the send buffer, written through (sc1): what the part counter announces has to be in memory, not in this XCD's L2.  Base
            // of the buffer descriptor: the tile's plane and (rotated) column, wave-uniform; per thread a 32-bit byte offset (the
            // launcher checks that the packed layout of one slab stays below 4 GiB)
            int col = (int)(un * CB);
            if (rm.rot != 0) col = (col + rm.rot * (int)(plane + (unsigned)rm.a0)) & rm.mask;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(dst + (long long)plane * pk_plane + col), 0, (int)0xffffffffu, 0x00020000);
            const unsigned               thr = (unsigned)pk_thr;
#pragma unroll
            for (int k = 0; k < EY; ++k)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(zy_u32x4, v[k]), rs, (int)((thr + pk_off(k)) * 16u), 0, DFFT_ZY_SIG_AUX);
        } else if constexpr (ROWS_PRODUCE && PACK) {
            V* op = dst + pk_base(plane, un);  // the send buffer: not read again by this device, streamed out
#pragma unroll
            for (int k = 0; k < EY; ++k) gstore<true>(op + pk_off(k), v[k]);
        } else if constexpr (ROWS_PRODUCE) {
            V* op = w + (long long)plane * w_plane + (long long)jy * DFFT_ZY_WP + un * CB + cy;
#pragma unroll
            for (int k = 0; k < EY; ++k) op[(long long)(TY * k) * DFFT_ZY_WP] = v[k];
        } else {
            const __amdgpu_buffer_rsrc_t rs = wrsrc(plane);
#pragma unroll
            for (int k = 0; k < EY; ++k) {
                const unsigned elem = (unsigned)((jy + TY * k) * DFFT_ZY_WP + un * CB + cy);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(zy_u32x4, v[k]), rs, (int)(elem * 16u), 0, 16 /* sc1 */);
            }
        }
    };
    auto is_rows = [&](unsigned kind) { return (kind == PROD) == ROWS_PRODUCE; };
    auto load_unit = [&](const Item& it, V* d) {
        if (is_rows(it.kind)) load_rows(it.plane, it.unit, d);
        else load_cols(it.plane, it.unit, d);
    };
    auto publish = [&](unsigned plane) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave: the unit's results have left this CU
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(&ctl->done[plane], 1u, __ATOMIC_RELAXED, DFFT_ZY_AGENT);
    };
    auto compute_unit = [&](const Item& it, V* v) {
        if (is_rows(it.kind)) {
            run_stages<V, PZ, 0, SIGN, 1, true, true, TWMZ, TWPZ>(v, twzp, lds_row, jz, 0);
        } else {
            __syncthreads();  // the LDS rows of an earlier row unit are no longer read
            run_stages<V, PY, 0, SIGN, CB, false, false, TWMY, TWPY, 1, 1, NWY>(v, twyp, lds, jy, cy);
        }
    };
    auto store_unit = [&](const Item& it, const V* v) {
        if (is_rows(it.kind)) {
            store_rows(it.plane, it.unit, v);
        } else {
            store_cols(it.plane, it.unit, v);
            __syncthreads();  // the tile is no longer read when the next unit scatters
        }
    };

    V v[E], vn[E];
    // tickets are taken two items ahead, so that the atomic's round trip overlaps a whole unit of work
    // A launch takes total + 2 * grid tickets in all.  A first ticket outside that range means the control block's counter is out
    // of step with the host's ticket_base (an earlier launch gave up before its workgroups had taken their last tickets): every
    // item would decode as NONE and the launch would "finish" without computing -- refuse loudly instead.  So does every launch
    // behind a failed one (sticky error word).
    unsigned first = take();
    if (tid == 0) {
        if (__hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, DFFT_ZY_AGENT) != 0u) {
            first = 0xffffffffu;
        } else if (first >= total + 2u * gridDim.x) {
            raise(ZY_ERR_DESYNC);
            first = 0xffffffffu;
        }
    }
    first = share(first);
    if (first == 0xffffffffu) return;
    Item cur = decode(first);
    Item nxt = decode(share(take()));
    if (cur.kind != NONE) {
        if (!ready(cur, true)) return;
        load_unit(cur, v);
    }
    if constexpr (!LAZY) {
        while (cur.ticket < total) {
            const unsigned t2 = take();
            bool           loaded = false;
            if (nxt.kind != NONE && ready(nxt, false)) {  // prefetch only what may be read already
                load_unit(nxt, vn);
                loaded = true;
            }
            if (cur.kind != NONE) {
                compute_unit(cur, v);
                store_unit(cur, v);
                if (cur.kind == PROD) publish(cur.plane);
            }
            const Item nn = decode(share(t2));
            if (nxt.kind != NONE && !loaded) {
                if (!ready(nxt, true)) return;
                load_unit(nxt, v);
            } else if (loaded) {
#pragma unroll
                for (int k = 0; k < E; ++k) v[k] = vn[k];
            }
            cur = nxt;
            nxt = nn;
        }
    } else {
        // One quiet point per unit, between its arithmetic and its stores: there everything the workgroup has in flight -- the
        // previous unit's stores, the next unit's loads, the ticket atomic -- has had the unit's exchanges to complete, so the
        // (all-or-nothing) vmcnt wait is short; the previous producer unit is published, the ticket read, the dependency of the
        // unit after next polled.  After the stores nothing waits, so they drain underneath the next unit.
        constexpr unsigned NOPLANE = 0xffffffffu;
        unsigned           pend = NOPLANE;  // plane of a producer unit whose results are stored but not yet published
        unsigned           pend_sig = NOPLANE;  // SIG: part of a column unit whose results are stored but not yet counted
        auto               flush = [&]() {
            if (pend != NOPLANE) publish(pend);
            pend = NOPLANE;
            if constexpr (SIG) {
                if (pend_sig != NOPLANE) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave: the unit's results are in memory
                    __syncthreads();
                    if (tid == 0) __hip_atomic_fetch_add(&part_done[pend_sig], 1u, __ATOMIC_RELAXED, DFFT_ZY_AGENT);
                }
                pend_sig = NOPLANE;
            }
        };
        bool nxt_ok = false;  // the dependency of nxt is known to be satisfied (polled at an earlier quiet point)
        while (cur.ticket < total) {
            // the products of the stored twiddle powers (w^3 = w^1 w^2, ...) are loop-invariant, and hoisted out of the unit loop they
            // would occupy the registers the power form exists to save (this variant then spills them): keep them per unit
#pragma unroll
            for (int i = 0; i < TWNZ; ++i) asm volatile("" : "+v"(twzr[i].x), "+v"(twzr[i].y));
#pragma unroll
            for (int i = 0; i < TWNY; ++i) asm volatile("" : "+v"(twyr[i].x), "+v"(twyr[i].y));
            const unsigned t2 = take();
            bool           loaded = false;
            if (nxt.kind != NONE && (nxt_ok || ready(nxt, false))) {
                load_unit(nxt, vn);
                loaded = true;
            }
            Item nn{0u, NONE, 0u, 0u};
            bool nn_ok = false;
            if (cur.kind != NONE) {
                compute_unit(cur, v);
                flush();
                if (loaded) {
#pragma unroll
                    for (int k = 0; k < E; ++k) pin_loaded(vn[k]);
                }
                nn = decode(share(t2));
                nn_ok = nn.kind != CONS || ready(nn, false);
                store_unit(cur, v);
                if (cur.kind == PROD) pend = cur.plane;
                else if constexpr (SIG) pend_sig = (cur.plane - plane0) / part_planes;
            } else {
                nn = decode(share(t2));
            }
            if (nxt.kind != NONE && !loaded) {
                flush();  // never wait for other workgroups while holding an unpublished unit
                if (!ready(nxt, true)) return;
                load_unit(nxt, v);
            } else if (loaded) {
#pragma unroll
                for (int k = 0; k < E; ++k) v[k] = vn[k];
            }
            cur = nxt;
            nxt = nn;
            nxt_ok = nn_ok;
        }
        flush();
    }
}

template <class PZ, class PY, int DIR, bool PACK, bool LAZY = false, int SIGN = DIR, bool SIG = false> hipError_t launch_zy_t(const ZyLaunch& L, hipStream_t stream) {
    constexpr int    CB = ZyTile<PY, PACK>::CB, THREADS = ZyTile<PY, PACK>::THREADS, GR = THREADS / PZ::T;
    constexpr unsigned UA = DIR > 0 ? (unsigned)(PY::N / GR) : (unsigned)(PZ::N / CB);  // producer units per plane, as in the kernel
    constexpr size_t ROW_BYTES = (size_t)GR * (PZ::N + PZ::N / 8) * sizeof(double2), COL_BYTES = (size_t)PY::N * CB * sizeof(double2);
    constexpr size_t TW_BYTES = (TwTotal<PZ, true>::value > 16 ? (size_t)PZ::N * sizeof(double2) : 0) + (TwTotal<PY, true>::value > 16 ? (size_t)PY::N * sizeof(double2) : 0);
    constexpr size_t LDS_BYTES = 64 + TW_BYTES + (ROW_BYTES > COL_BYTES ? ROW_BYTES : COL_BYTES);
    static_assert(LDS_BYTES <= 160 * 1024, "a unit and the twiddle tables must fit the LDS of a CU");
    auto             kern = zy_chunk_kernel<PZ, PY, DIR, PACK, LAZY, SIGN, SIG>;
    static std::atomic<bool> attr_set[64];
    static std::mutex        setup_mutex;
    int                      dev = 0;
    hipError_t               e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lk(setup_mutex);
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set[dev].store(true, std::memory_order_release);
    }
    constexpr int PK_GRAIN = (PY::T == 64 && (LAZY || PY::N == 768)) ? 32 : PY::T;  // (PK2 in the kernel)
    if (SIG && (!L.part_done || L.part_planes <= 0)) return hipErrorInvalidValue;
    if (PACK && (L.pk.blk <= 0 || L.pk.blk % PK_GRAIN != 0 || L.pk.last_delta != 0)) return hipErrorInvalidValue;
    // one workgroup per CU (the shape measured in round 2; a second one per CU gained nothing -- nor do two or three of the
    // 256-thread workgroups of 256-point Y axes, profiles/r03/experiments/variant_ab_256.log)
    const long long grid = zy_grid();
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(THREADS), LDS_BYTES, stream, (const double2*)L.src, (double2*)L.w, (double2*)L.dst, L.ctl,
                       (const double2*)L.twz, (const double2*)L.twy, L.src_plane, L.w_plane, L.dst_plane, (unsigned)L.plane0, (unsigned)L.nplanes,
                       (unsigned)L.chunk, L.ticket_base, L.done_base, L.pk, L.pk_plane, L.rot, L.err_host, L.spin_polls, UA + (L.fault ? 1u : 0u), L.part_done, (unsigned)L.part_planes
#if DFFT_ZY_ROW_PITCH
                       , (unsigned)L.w_pitch
#endif
    );
    return hipGetLastError();
}

using P256 = Plan<256, 8, 8, 8, 4>;
using P512 = Plan<512, 8, 8, 8, 8>;
// the library's column plan for 768 points (dfft_plans.h), so that the arithmetic is the two-launch path's: 24 points x 32 threads
// (256-thread units) or, -DDFFT_768_E12=1, 12 points x 64 threads (512-thread units like the other plane shapes)
#if DFFT_768_E12
using P768 = Plan<768, 12, 4, 4, 4, 4, 3>;
constexpr int ZY_E768 = 12;
#else
using P768 = Plan<768, 24, 8, 8, 4, 3>;
constexpr int ZY_E768 = 24;
#endif

// host-side mirror of the kernel's geometry: threads per workgroup, columns per tile, rows per row unit
struct ZyGeom {
    int threads, cb, gr;
};
ZyGeom zy_geom(int n1, int n2, int packed) {
    const int ey = n1 == 768 ? ZY_E768 : 8, ty = n1 / ey, tz = n2 / 8;
    const int cb = (!packed && ty < 64) ? 8 * (64 / ty) : 8;  // ZyTile
    if (n1 == 768) return ZyGeom{8 * ty, 8, 8 * ty / tz};      // (a two-line tile of 768 points would not fit the LDS)
    return ZyGeom{cb * ty, cb, cb * ty / tz};
}

}  // namespace

// The exchange stream's side of SIG launches: one wave that returns when the part's counter has reached `target` (wrapping compare),
// the stage has given up (sticky error word of the control block), or the time limit has passed (then it raises the plan's error word
// itself: the exchange behind it ships an incomplete part, and the host reports the execute as invalid like any other failure of the stage).
namespace {
__global__ void zy_part_wait_kernel(const unsigned* ctr, unsigned target, const ZyCtl* ctl, unsigned* err_host, unsigned long long limit_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();  // 100 MHz
    while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, DFFT_ZY_AGENT) - target) < 0) {
        if (__hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, DFFT_ZY_AGENT) != 0u) return;
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > limit_ticks) {
            __hip_atomic_store(err_host, (unsigned)ZY_ERR_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
    }
}
}  // namespace
hipError_t launch_zy_part_wait(const unsigned* ctr, unsigned target, const ZyCtl* ctl, unsigned* err_host, hipStream_t stream) {
    static const unsigned long long limit = [] {
        const char*  e = getenv("DFFT_ZY_PART_TIMEOUT_S");
        const double s = e && atof(e) > 0 ? atof(e) : 30.0;
        return (unsigned long long)(s * 1e8);
    }();
    (void)hipGetLastError();
    hipLaunchKernelGGL(zy_part_wait_kernel, dim3(1), dim3(64), 0, stream, ctr, target, ctl, err_host, limit);
    return hipGetLastError();
}

// fp64; Y axis of 256, 512 or (round 5) 768 points, Z axis of 256 or 512 points
bool zy_supported(int dtype, int n1, int n2) { return dtype == F64 && (n1 == 256 || n1 == 512 || (n1 == 768 && n2 == 512)) && (n2 == 256 || n2 == 512); }
// threads that share one column FFT of the Y axis: a destination block of the packed layout must be a whole number of them
int zy_col_threads(int n1, int lazy) { return (n1 == 768 || (n1 == 512 && lazy)) ? 32 : n1 / 8; }  // (64 threads per column -- 512 (lazy form), 768 points: destination blocks of whole 32 rows, PK2)

// workgroups per launch: one per CU.  Every workgroup takes tickets until it sees one past the end, holding two ahead, so a launch
// advances the ticket counter by its item count + 2 per workgroup: zy_tickets() is what the host adds to its running ticket base.
long long zy_grid() { return device_info().cus; }
unsigned  zy_units_per_plane(int n1, int n2, int dir, int packed, unsigned* producers) {
    const ZyGeom   g = zy_geom(n1, n2, packed);
    const unsigned uz = (unsigned)(n1 / g.gr), uy = (unsigned)(n2 / g.cb);
    if (producers) *producers = dir > 0 ? uz : uy;
    return uz + uy;
}
unsigned zy_tickets(int n1, int n2, int dir, int packed, long long nplanes, long long chunk) {
    const unsigned nchunks = (unsigned)((nplanes + chunk - 1) / chunk);
    return nchunks * (unsigned)chunk * zy_units_per_plane(n1, n2, dir, packed, nullptr) + 2u * (unsigned)zy_grid();
}

hipError_t launch_zy(const ZyLaunch& L, hipStream_t stream) {
    if (!zy_supported(L.dtype, L.n1, L.n2) || L.nplanes <= 0 || L.plane0 + L.nplanes > ZY_MAX_PLANES || L.chunk <= 0 || !L.err_host || L.spin_polls == 0)
        return hipErrorInvalidValue;
#if DFFT_ZY_ROW_PITCH
    if (L.w_pitch < L.n2 || (long long)L.n1 * L.w_pitch * 16 >= (1ll << 31)) return hipErrorInvalidValue;  // (32-bit buffer offsets inside a plane)
#endif
#define DFFT_ZY_CASE(NZ, NY, PZ_, PY_)                                                                                           \
    if (L.n2 == NZ && L.n1 == NY) {                                                                                             \
        if (L.lazy && !L.packed && L.dir > 0 && L.sign < 0) return launch_zy_t<PZ_, PY_, +1, false, true, -1>(L, stream);                         \
        if (L.sign != 0 && L.sign != L.dir) return hipErrorInvalidValue;                                                                        \
        if (L.part_done && !(L.lazy && L.packed && L.dir > 0)) return hipErrorInvalidValue;                                                     \
        if (L.part_done) return launch_zy_t<PZ_, PY_, +1, true, true, +1, true>(L, stream);                                                     \
        if (L.lazy && L.packed) return L.dir > 0 ? launch_zy_t<PZ_, PY_, +1, true, true>(L, stream) : launch_zy_t<PZ_, PY_, -1, true, true>(L, stream); \
        if (L.lazy) return L.dir > 0 ? launch_zy_t<PZ_, PY_, +1, false, true>(L, stream) : launch_zy_t<PZ_, PY_, -1, false, true>(L, stream); \
        if (L.packed) return L.dir > 0 ? launch_zy_t<PZ_, PY_, +1, true>(L, stream) : launch_zy_t<PZ_, PY_, -1, true>(L, stream); \
        return L.dir > 0 ? launch_zy_t<PZ_, PY_, +1, false>(L, stream) : launch_zy_t<PZ_, PY_, -1, false>(L, stream);            \
    }
    DFFT_ZY_CASE(512, 512, P512, P512)
    DFFT_ZY_CASE(256, 256, P256, P256)
    DFFT_ZY_CASE(512, 256, P512, P256)
    DFFT_ZY_CASE(256, 512, P256, P512)
    DFFT_ZY_CASE(512, 768, P512, P768)
#undef DFFT_ZY_CASE
    return hipErrorInvalidValue;
}

}  // namespace dfft
