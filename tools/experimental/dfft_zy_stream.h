// dfft_zy_stream.h -- EXPERIMENT (not linked into the library; tools/zy_stream.hip is the harness that verifies it against the
// library kernels and times it).  First GPU run (profiles/r02/experiments/zy_stream_first_run.log): every variant bit-identical to
// the library; the interleaved orders are all slower than its phase-separated chunks (1.52 ms at best against 1.37 ms), the phase
// order CHUNK = 64 with the sc1 hand-off is 2 % faster (1.337 ms: the launch boundaries), fences per bundle cost 0.15-0.2 ms.
//
// t0 (batched 2D YZ FFT of every owned plane; reference fftZY, /root/reference/3dmpifft_opt/include/fft_mpi_3d_api.cpp:466-522)
// as ONE persistent launch in which the Z rows of plane p + LAG and the Y columns of plane p are in flight AT THE SAME TIME.
//
// Why (profiles/r02/README.md): the library's t0 alternates two launches per 256 MiB cache chunk.  The Z
// launch carries all of the stage's HBM traffic (input read + write-back of the previous chunk), the Y launch is a pure
// Infinity-Cache pass; both move ~6.1-6.4 TB/s across the XCD <-> fabric boundary, whose measured ceiling is 7.1-7.6 TB/s, and
// every one of the 16 launch boundaries costs ~4.5 us.  Running the two kinds of work concurrently on two streams was slower
// (profiles/r01/experiments/kbench_overlap.log) because a stream can only overlap whole chunks -- two chunks in flight do not
// fit the cache -- and because of the event traffic between the streams.  Here the unit of dependency is ONE PLANE:
//   * work items are handed out by a global ticket counter in a fixed order: for block b = 0, 1, ... the Z units of plane b
//     (GR rows each, one row per wavefront) interleaved with the Y units of plane b - LAG (one 128-byte-wide column tile each);
//   * a Z unit publishes its rows (write-through stores, drained, then one relaxed agent-scope increment of done[plane]);
//   * a Y unit starts when done[plane] has reached the number of Z units of its plane -- which happened ~LAG * (UZ + UY)
//     tickets ago, so the poll practically never waits -- and transforms its tile in place.
// The Infinity-Cache working set is LAG planes (LAG = 16: 64 MiB) instead of a 256 MiB chunk, HBM reads (Z) and cache-only
// traffic (Y) overlap all the time, and there are no launch boundaries inside t0.  Expected from the boundary ceiling:
// 8.6 GB / 7.5 TB/s = 1.15 ms against 1.37 ms -- NOT what the first run showed: the mix is slower than the phases.
//
// Visibility (MI355X_MICROARCH.md, inter-workgroup hand-off rules): the producer and the consumer of a row are in general on
// different XCDs, whose L2s are not coherent with each other, and a CU's L1 is never refreshed by other CUs' stores.
//   HANDOFF 0: rows stored with sc1 (write-through) 16-byte buffer stores, every storing wave drains (s_waitcnt vmcnt(0)),
//              __syncthreads, ONE relaxed agent-scope atomic; the consumer polls that word and reads the tile with sc1 loads
//              (no fences at all -- guide form "sc1 stores and loads on both sides").
//   HANDOFF 1: plain stores, drain, __syncthreads, lane 0: agent-scope release fence, then the atomic; the consumer: poll,
//              agent-scope acquire fence (one lane), __syncthreads, plain loads.
// No stale line can sit in the consumer XCD's L2: within one launch a W line is read only after the Z unit of this launch has
// written it, and the L2s are invalidated at the launch boundary.
// Deadlock freedom: tickets are taken in order by RUNNING workgroups only; a Z unit never waits; a Y unit waits only for Z units
// with smaller tickets, and a workgroup never sits on an unprocessed item while it waits (the next item is prefetched only if
// its dependency is already satisfied, otherwise the current item is finished first).  Every spin is bounded by the wall clock;
// a time-out sets ctl->error and every workgroup leaves (the host then falls back to the two-launch path).
#pragma once
#include "dfft_fft_impl.h"

namespace dfft {

struct alignas(128) ZyCtl {
    unsigned ticket;
    unsigned pad0[31];
    unsigned error;
    unsigned pad1[31];
    unsigned waits;       // statistics: Y units that found their plane unfinished at the first poll
    unsigned pad2[31];
    unsigned done[4096];  // per plane: Z units completed
};
enum { ZY_ERR_WAIT = 1 };

struct ZyCfgDefault {
    static constexpr int      THREADS = 512;
    static constexpr int      HANDOFF = 0;        // 0: sc1 stores + sc1 loads, 1: plain + release / acquire fences
    static constexpr bool     PREFETCH = true;    // load the next item before processing the current one
    static constexpr bool     FINE = true;        // tickets alternate Z / Y units (false: all Z units of a block, then all Y units)
    static constexpr bool     DYNAMIC = true;     // tickets from a global atomic counter (self-balancing); false: static rotation --
                                                  // workgroup g takes item m * G + (g + m) % G at its step m (no atomic, no broadcast;
                                                  // needs every workgroup of the grid resident, which the persistent grid guarantees)
    static constexpr bool     IN_NT = true;       // streamed input
    static constexpr bool     OUT_NT = false;     // Y results: plain stores (the X pass finds the tail in the cache)
    static constexpr bool     MATH = true;        // false: data movement only (measurement builds)
    static constexpr int      BUNDLE = 1;         // units per ticket: a workgroup takes BUNDLE consecutive units of one plane and pays the
                                                  // hand-off cost (drain + release fence + counter / poll + acquire fence) once per bundle
    static constexpr int      CHUNK = 0;          // > 0: phase order -- all Z bundles of CHUNK planes, then all Y bundles of the same planes
                                                  // (the library's cache chunks without the launch boundaries; `lag` is ignored)
    static constexpr unsigned TIMEOUT_TICKS = 20u * 1000u * 100u;  // 20 ms of the 100 MHz wall clock
};

#define DFFT_ZY_AGENT __HIP_MEMORY_SCOPE_AGENT
typedef unsigned zy_u32x4 __attribute__((ext_vector_type(4)));

// in : [plane][N1][N2] natural layout, plane stride in_plane elements
// w  : [plane][N1][N2] rows N2 apart, planes w_plane elements apart (the plan's padded work buffer); Z writes it, Y works in place
template <class V, class PZ, class PY, int DIR, class Cfg>
__global__ void __attribute__((amdgpu_flat_work_group_size(Cfg::THREADS, Cfg::THREADS), amdgpu_waves_per_eu(1)))
zy_stream_kernel(const V* in, V* w, ZyCtl* ctl, const V* __restrict__ twz, const V* __restrict__ twy, long long in_plane, long long w_plane,
                 unsigned nplanes, unsigned lag) {
    static_assert(sizeof(V) == 16, "experiment: 16-byte elements (fp64 complex)");
    constexpr int THREADS = Cfg::THREADS;
    constexpr int N2 = PZ::N, N1 = PY::N;
    constexpr int EZ = PZ::E, TZ = PZ::T, EY = PY::E, TY = PY::T;
    static_assert(EZ == EY, "one register set serves both item kinds");
    constexpr int E = EZ;
    static_assert(THREADS % TZ == 0 && TZ <= 64 && 64 % TZ == 0, "rows: one FFT inside one wavefront");
    constexpr int GR = THREADS / TZ;  // rows per Z unit
    static_assert(THREADS % TY == 0, "columns: the workgroup is one tile");
    constexpr int CB = THREADS / TY;  // columns per Y unit
    static_assert(CB * sizeof(V) == 128, "column tiles of one cache line");
    static_assert(N1 % GR == 0 && N2 % CB == 0, "plane must split into whole units");
    constexpr unsigned UZ = N1 / GR, UY = N2 / CB, B = UZ + UY;
    static_assert(!Cfg::FINE || UZ == UY, "alternating tickets need as many Z as Y units per plane");
    constexpr bool     TWPOW = true;
    constexpr int      ROW_LDS = N2 + N2 / 8;  // padded row (lds_index<1, true>)
    constexpr unsigned LIMIT = Cfg::TIMEOUT_TICKS;

    extern __shared__ __attribute__((aligned(16))) char dfft_smem[];
    unsigned* shw = reinterpret_cast<unsigned*>(dfft_smem);  // [0] ticket broadcast, [1] dependency state
    V*        lds = reinterpret_cast<V*>(dfft_smem + 64);

    const int tid = threadIdx.x;
    const int gz = tid / TZ, jz = tid % TZ;  // Z unit: row gz of the unit, butterfly id jz
    const int cy = tid % CB, jy = tid / CB;  // Y unit: column cy of the tile, butterfly id jy
    V*        lds_row = lds + gz * ROW_LDS;

    constexpr int TWNZ = TwTotal<PZ, TWPOW>::value, TWNY = TwTotal<PY, TWPOW>::value;
    V twzr[Cfg::MATH && TWNZ > 0 ? TWNZ : 1], twyr[Cfg::MATH && TWNY > 0 ? TWNY : 1];
    if constexpr (Cfg::MATH) {
        load_twiddles<V, PZ, 0, DIR, TWPOW>(twzr, twz, jz);
        load_twiddles<V, PY, 0, DIR, TWPOW>(twyr, twy, jy);
    }

    constexpr unsigned K = Cfg::BUNDLE;
    static_assert(UZ % K == 0 && UY % K == 0, "bundles must tile a plane's units");
    constexpr unsigned ZB = UZ / K, YB = UY / K, BB = ZB + YB;  // bundles per plane
    static_assert(!Cfg::FINE || Cfg::CHUNK > 0 || ZB == YB, "alternating tickets need as many Z as Y bundles per plane");
    constexpr unsigned CH = Cfg::CHUNK > 0 ? (unsigned)Cfg::CHUNK : 1u;
    const unsigned nchunks = (nplanes + CH - 1) / CH;
    const unsigned total = Cfg::CHUNK > 0 ? nchunks * CH * BB : (nplanes + lag) * BB;  // tickets (= bundles, some of them empty)
    enum { NONE = 0, ZU = 1, YU = 2 };
    struct Item {  // one bundle: units [unit, unit + K) of `plane`
        unsigned ticket, kind, plane, unit;
    };
    auto decode = [&](unsigned t) -> Item {
        Item it{t, NONE, 0u, 0u};
        if (t >= total) return it;
        if constexpr (Cfg::CHUNK > 0) {
            const unsigned c = t / (CH * BB), r = t - c * (CH * BB);
            if (r < CH * ZB) {
                const unsigned pl = c * CH + r / ZB;
                if (pl < nplanes) it = Item{t, ZU, pl, (r % ZB) * K};
            } else {
                const unsigned r2 = r - CH * ZB, pl = c * CH + r2 / YB;
                if (pl < nplanes) it = Item{t, YU, pl, (r2 % YB) * K};
            }
            return it;
        }
        const unsigned b = t / BB, r = t - b * BB;
        bool           z;
        unsigned       u;
        if constexpr (Cfg::FINE) {
            z = (r & 1u) == 0u;
            u = r >> 1;
        } else {
            z = r < ZB;
            u = z ? r : r - ZB;
        }
        if (z) {
            if (b < nplanes) it = Item{t, ZU, b, u * K};
        } else if (b >= lag) {
            it = Item{t, YU, b - lag, u * K};
        }
        return it;
    };
    unsigned step = 0;  // static ticket order only
    auto share = [&](unsigned value_of_thread0) -> unsigned {  // broadcast a value held by thread 0
        if constexpr (!Cfg::DYNAMIC) return value_of_thread0;  // every thread computed it
        if (tid == 0) shw[0] = value_of_thread0;
        __syncthreads();
        const unsigned t = shw[0];
        __syncthreads();
        return t;
    };
    auto take = [&]() -> unsigned {  // thread 0: the next ticket (the atomic's latency is hidden behind the caller's work)
        if constexpr (!Cfg::DYNAMIC) {
            const unsigned t = step * gridDim.x + (blockIdx.x + step) % gridDim.x;
            ++step;
            return t;
        }
        return tid == 0 ? __hip_atomic_fetch_add(&ctl->ticket, 1u, __ATOMIC_RELAXED, DFFT_ZY_AGENT) : 0u;
    };
    // dependency of a Y bundle: all Z units of its plane have published.  wait = false: one poll only.
    auto ready = [&](const Item& it, bool wait) -> bool {
        if (it.kind != YU) return true;
        if (tid == 0) {
            unsigned ok = __hip_atomic_load(&ctl->done[it.plane], __ATOMIC_RELAXED, DFFT_ZY_AGENT) >= UZ ? 1u : 0u;
            if (!ok && wait) {
                __hip_atomic_fetch_add(&ctl->waits, 1u, __ATOMIC_RELAXED, DFFT_ZY_AGENT);
                const unsigned long long t0 = wall_clock64();
                for (;;) {
                    __builtin_amdgcn_s_sleep(1);
                    if (__hip_atomic_load(&ctl->done[it.plane], __ATOMIC_RELAXED, DFFT_ZY_AGENT) >= UZ) {
                        ok = 1u;
                        break;
                    }
                    if (__hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, DFFT_ZY_AGENT) != 0u) break;
                    if (wall_clock64() - t0 > LIMIT) {
                        __hip_atomic_store(&ctl->error, (unsigned)ZY_ERR_WAIT, __ATOMIC_RELAXED, DFFT_ZY_AGENT);
                        break;
                    }
                }
            }
            if (ok && Cfg::HANDOFF == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // this CU's L1 drops its lines
            shw[1] = ok;
        }
        __syncthreads();
        const bool ok = shw[1] != 0u;
        __syncthreads();
        return ok;
    };
    auto wrsrc = [&](unsigned plane) {  // buffer descriptor of one plane of w (offsets inside a plane fit 32 bits)
        return __builtin_amdgcn_make_buffer_rsrc((void*)(w + (long long)plane * w_plane), 0, (int)((size_t)N1 * N2 * sizeof(V)), 0x00020000);
    };
    // unit `un` of the bundle's plane
    auto load_unit = [&](unsigned kind, unsigned plane, unsigned un, V* dst) {
        if (kind == ZU) {
            const V* ip = in + (long long)plane * in_plane + (long long)(un * GR + gz) * N2 + jz;
#pragma unroll
            for (int k = 0; k < E; ++k) dst[k] = gload<Cfg::IN_NT>(ip + TZ * k);
        } else if (kind == YU) {
            if constexpr (Cfg::HANDOFF == 0) {
                const __amdgpu_buffer_rsrc_t rs = wrsrc(plane);
#pragma unroll
                for (int k = 0; k < E; ++k) {
                    const unsigned elem = (unsigned)((jy + TY * k) * N2 + un * CB + cy);
                    dst[k] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(elem * 16u), 0, 16 /* sc1 */));
                }
            } else {
                const V* ip = w + (long long)plane * w_plane + (long long)jy * N2 + un * CB + cy;
#pragma unroll
                for (int k = 0; k < E; ++k) dst[k] = ip[(long long)(TY * k) * N2];
            }
        }
    };
    // transform + store one unit; `last`: the bundle's last unit publishes (Z)
    auto process_unit = [&](unsigned kind, unsigned plane, unsigned un, bool last, V* v) {
        if (kind == ZU) {
            if constexpr (Cfg::MATH) run_stages<V, PZ, 0, DIR, 1, true, true, TW_REG, TWPOW>(v, twzr, lds_row, jz, 0);
            if constexpr (Cfg::HANDOFF == 0) {
                const __amdgpu_buffer_rsrc_t rs = wrsrc(plane);
#pragma unroll
                for (int k = 0; k < E; ++k) {
                    const unsigned elem = (unsigned)((un * GR + gz) * N2 + jz + TZ * k);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(zy_u32x4, v[k]), rs, (int)(elem * 16u), 0, 16 /* sc1 */);
                }
            } else {
                V* op = w + (long long)plane * w_plane + (long long)(un * GR + gz) * N2 + jz;
#pragma unroll
                for (int k = 0; k < E; ++k) op[TZ * k] = v[k];
            }
            if (last) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave: the bundle's rows have left this CU
                __syncthreads();
                if (tid == 0) {
                    if constexpr (Cfg::HANDOFF == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    __hip_atomic_fetch_add(&ctl->done[plane], K, __ATOMIC_RELAXED, DFFT_ZY_AGENT);
                }
            }
        } else if (kind == YU) {
            if constexpr (Cfg::MATH) {
                __syncthreads();  // the LDS rows of an earlier Z unit are no longer read
                run_stages<V, PY, 0, DIR, CB, false, false, TW_REG, TWPOW>(v, twyr, lds, jy, cy);
            }
            V* op = w + (long long)plane * w_plane + (long long)jy * N2 + un * CB + cy;
#pragma unroll
            for (int k = 0; k < E; ++k) gstore<Cfg::OUT_NT>(op + (long long)(TY * k) * N2, v[k]);
            if constexpr (Cfg::MATH) __syncthreads();  // the tile is no longer read when the next unit scatters
        }
    };

    V    v[E], vn[Cfg::PREFETCH ? E : 1];
    // tickets are taken two bundles ahead, so that the atomic's round trip overlaps a whole bundle of work
    Item cur = decode(share(take()));
    Item nxt = decode(share(take()));
    if (cur.kind != NONE) {
        if (!ready(cur, true)) return;
        load_unit(cur.kind, cur.plane, cur.unit, v);
    }
    while (cur.ticket < total) {
        const unsigned t2 = take();
        // units 0 .. K-2 of the bundle: the next unit belongs to the same bundle, nothing to check
#pragma unroll 1
        for (unsigned i = 0; i + 1 < K; ++i) {
            if (cur.kind == NONE) break;
            if constexpr (Cfg::PREFETCH) load_unit(cur.kind, cur.plane, cur.unit + i + 1, vn);
            process_unit(cur.kind, cur.plane, cur.unit + i, false, v);
            if constexpr (Cfg::PREFETCH) {
#pragma unroll
                for (int k = 0; k < E; ++k) v[k] = vn[k];
            } else {
                load_unit(cur.kind, cur.plane, cur.unit + i + 1, v);
            }
        }
        // last unit of the bundle: the next unit is the first of the next bundle (prefetched only if its dependency is satisfied)
        bool loaded = false;
        if constexpr (Cfg::PREFETCH) {
            if (nxt.kind != NONE && ready(nxt, false)) {
                load_unit(nxt.kind, nxt.plane, nxt.unit, vn);
                loaded = true;
            }
        }
        if (cur.kind != NONE) process_unit(cur.kind, cur.plane, cur.unit + K - 1, true, v);
        const Item nn = decode(share(t2));
        if (nxt.kind != NONE && !loaded) {
            if (!ready(nxt, true)) return;
            load_unit(nxt.kind, nxt.plane, nxt.unit, v);
        } else if constexpr (Cfg::PREFETCH) {
            if (loaded) {
#pragma unroll
                for (int k = 0; k < E; ++k) v[k] = vn[k];
            }
        }
        cur = nxt;
        nxt = nn;
    }
}

template <class V, class PZ, class PY, class Cfg> struct ZyGeom {
    static constexpr int    GR = Cfg::THREADS / PZ::T;
    static constexpr int    CB = Cfg::THREADS / PY::T;
    static constexpr size_t ROW_BYTES = (size_t)GR * (PZ::N + PZ::N / 8) * sizeof(V);
    static constexpr size_t COL_BYTES = (size_t)PY::N * CB * sizeof(V);
    static constexpr size_t LDS_BYTES = 64 + (ROW_BYTES > COL_BYTES ? ROW_BYTES : COL_BYTES);
};

}  // namespace dfft
