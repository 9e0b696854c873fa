// dfft_exchange.cpp -- t2, the personalised all-to-all that re-slabs X -> Y.
//
// Replaces slabAlltoall (fft_mpi_3d_api.cpp:610-672): the reference moves chunk (src -> dst) with hipMemcpyPeerAsync
// inside a process and with MPI_Isend/Irecv on device pointers (UCX) between processes.  Here:
//   RCCL  communicator : one grouped ncclSend/ncclRecv per peer on the given stream (xGMI point-to-point: each
//                        (src,dst) pair has its own link, so the whole exchange is bounded by the pair chunk
//                        S*N/P^2 over ~153 GB/s), self chunk by device copy.
//   LOCAL communicator : P device-threads of one process; peer copies between the plans' buffers with the same
//                        barrier discipline as the reference's `#pragma omp barrier`s (fft_mpi_3d_api.cpp:190,195).
//                        Also used with P virtual devices on one physical GPU for single-GPU parity tests.
// Because the Y pass writes the packed [dest][x][y_dest][N2] layout directly, every message is one contiguous block on
// both sides, and so is any sub-range of its X planes: comm_exchange_part moves planes [k*cp, (k+1)*cp) of every
// source slab, which lets the plan pipeline t2 behind the plane-chunked t0.
#include <rccl/rccl.h>
#include <unistd.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <map>
#include <mutex>
#include <string>

#include "dfft_internal.h"

struct dfft_comm_s {
    int kind = 0;  // 0 local (threads of one process), 1 rccl, 2 ipc (one process per device, host-synchronised peer copies)
    int P = 1;
    // local
    std::mutex              m;
    std::condition_variable cv;
    int                     arrived = 0;
    unsigned long           generation = 0;
    bool                    barrier_broken = false;  // a device thread gave up waiting in comm_thread_barrier: every later barrier fails at once
    int                     agree_in = 0;       // comm_agree_max: the devices' contributions of the round in progress
    int*                    agree_dev = nullptr;  // rccl: two device words for the one-element all-reduce
    // IPC communicators: receive buffers and their registrations (the peers' mappings, the flag words) are kept for the life of the
    // communicator and handed to the next plan that asks with the same key (comm_recv_alloc) -- see there for why
    struct PoolEntry {
        std::string   key;  // role of the buffer in a plan ("b1", "nat", "rb"): any unused entry of the role that is large enough is reused
        void*         buf;
        size_t        bytes;
        int           reg;
        bool          used;
        unsigned long stamp;  // pool_clock when the entry was last handed out or handed back (eviction order: least recent first)
    };
    std::vector<PoolEntry>       pool;
    std::map<void*, std::string> pending;  // fresh allocations of comm_recv_alloc that are not registered yet
    bool                         pooled = false;
    size_t                       pool_budget = 0;  // bytes the pool may hold before idle entries are evicted (DFFT_IPC_POOL_MB; 0 = no limit)
    unsigned long                pool_clock = 0;
    std::vector<std::pair<uintptr_t, size_t>> retired;  // address ranges that were exported once and freed: never exported again
    // Receive buffers by registration: every plan registers its receive buffer(s) in creation order, and plans are created
    // in the same order on every device thread / process, so registration r of device q is the buffer device q's r-th
    // exchange descriptor receives into (the reference shares ONE node_data[] between its forward and backward plan and
    // aliases them, fftSpeed3d_c2c.cpp:74,80; here any number of plans can be alive on one communicator).
    std::vector<std::vector<void*>> regs;      // [registration][device]
    std::vector<int>                next_reg;  // per device (local) / [rank] (ipc)
    std::vector<int>                devices;
    // ipc with stream-ordered synchronisation (kind 3): per registration, flag words shared like the receive buffers
    // (2*P words per device: ready[src], arrive[src]), a round counter, and a pinned host word a timed-out kernel sets
    std::vector<std::vector<unsigned long long*>> flag_regs;  // [registration][device]
    std::vector<unsigned long long>               seq;        // rounds issued per registration
    unsigned long long*                           err = nullptr;
    std::vector<hipStream_t>                      peer_streams;  // one helper stream + event per peer for the pushes
    std::vector<hipEvent_t>                       peer_events;
    hipEvent_t                                    fork_event = nullptr;
    // rccl
    ncclComm_t nccl = nullptr;
    int        rank = 0;
    int        device = 0;
};

namespace {
// Cross-process, stream-ordered synchronisation of the asynchronous IPC exchange: lane i publishes `seq` to sig[i] (a flag
// word in a peer's memory), then lane i waits until wait[i] (a flag word in this device's memory) has reached `seq`.
// Bounded: after DFFT_IPC_TIMEOUT_S seconds (default 20) the kernel gives up and reports through the pinned `err` word
// instead of hanging the device.
struct IpcSyncLists {
    unsigned long long*       sig[64];
    const unsigned long long* wait[64];
    int                       nsig, nwait;
};
__global__ void ipc_sync_kernel(IpcSyncLists L, unsigned long long seq, unsigned long long* err, unsigned long long limit_ticks) {
    const int i = threadIdx.x;
    if (i < L.nsig) __hip_atomic_store(L.sig[i], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (i < L.nwait) {
        // a round that has already timed out makes every later one give up at once: a dead peer costs ONE time limit, not one per
        // queued round (twelve overlapped executes queue ~120 rounds: round 4's "stall for minutes" behind a rank that had died)
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0ull) return;
        const unsigned long long t0 = wall_clock64();  // 100 MHz
        while (__hip_atomic_load(L.wait[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            __builtin_amdgcn_s_sleep(16);
            if (wall_clock64() - t0 > limit_ticks) {
                *err = seq;
                break;
            }
        }
    }
}
unsigned long long ipc_timeout_ticks() {  // wall_clock64 runs at 100 MHz
    static const unsigned long long t = [] {
        const char* e = std::getenv("DFFT_IPC_TIMEOUT_S");
        const double s = e && atof(e) > 0 ? atof(e) : 20.0;
        return (unsigned long long)(s * 1e8);
    }();
    return t;
}
}  // namespace

namespace dfft {

int comm_kind(dfft_comm_t c) { return c->kind; }
bool comm_is_async(dfft_comm_t c) { return c->kind == 1 || c->kind == 3; }
int comm_check(dfft_comm_t c) {
    if (c && c->kind == 3 && c->err && *(volatile unsigned long long*)c->err != 0)
        return fail(DFFT_ECOMM, "ipc exchange: a peer did not answer within the time limit (DFFT_IPC_TIMEOUT_S, default 20 s; round " + std::to_string(*c->err) + ")");
    return DFFT_OK;
}
int comm_size(dfft_comm_t c) { return c->P; }

int comm_thread_barrier(dfft_comm_t c) {
    if (c->kind == 2) return dfft_boot_barrier();  // processes instead of threads: the TCP rendezvous is the barrier
    if (c->kind != 0 || c->P <= 1) return DFFT_OK;
    // bounded: a device thread that never arrives -- it returned early from a collective call, or died -- must not hold
    // the others for ever.  DFFT_BOOT_TIMEOUT_S (default 180 s), the bound of the inter-process rendezvous; sticky once it has struck.
    static const double limit_s = [] {
        const char* e = std::getenv("DFFT_BOOT_TIMEOUT_S");
        return e && atof(e) > 0 ? atof(e) : 180.0;
    }();
    std::unique_lock<std::mutex> lk(c->m);
    if (c->barrier_broken) return fail(DFFT_ECOMM, "comm_thread_barrier: an earlier barrier of this communicator timed out");
    const unsigned long gen = c->generation;
    if (++c->arrived == c->P) {
        c->arrived = 0;
        ++c->generation;
        c->cv.notify_all();
    } else if (!c->cv.wait_for(lk, std::chrono::duration<double>(limit_s), [&] { return c->generation != gen || c->barrier_broken; }) || c->barrier_broken) {
        c->barrier_broken = true;
        c->cv.notify_all();
        return fail(DFFT_ECOMM, "comm_thread_barrier: not every device thread of the communicator arrived within " + std::to_string((int)limit_s) +
                                    " s (DFFT_BOOT_TIMEOUT_S): a collective call was skipped on one of them");
    }
    return DFFT_OK;
}

// Receive buffers of IPC communicators are POOLED (round 5).  Exporting a buffer to the peers (hipIpcGetMemHandle / OpenMemHandle),
// closing the mappings, freeing it and exporting the next allocation -- which the driver places on the just-freed memory -- is
// what a second generation of plans on one communicator does, and on this stack (ROCm 7, dmabuf IPC) it is not reliable:
// in tools/stall_hunt.py's four-process runs the second generation failed in 7 of 80 launches -- hipIpcGetMemHandle returning
// "invalid argument" (also when repeated), or, silently, a peer's fresh mapping still pointing at the memory of the buffer that
// had been there before, so that everything that peer pushed was lost and the plan's results were wrong for as long as it lived
// (profiles/r05/README.md section 1).  So a registered buffer is never unmapped or freed while its communicator lives: a plan that
// is destroyed hands buffer + registration back, and the next plan that asks with the same key (size, precision, device count,
// role) gets them -- no handle is created twice for one piece of memory, and re-planning costs no rendezvous round trips.  Plans
// are created and destroyed in the same order on every rank, so every rank makes the same choice.  DFFT_IPC_POOL=0 restores the
// export / unmap / free cycle per plan (A/B switch).
//
// Round 6: the pool is bounded and its hand-overs are checked.
//  * An idle entry serves ANY later plan that needs a buffer of the same role and at most its size (best fit), not only a plan of the
//    very same shape: an application that re-plans over many shapes on one communicator keeps as many buffers as it has plans alive at
//    once, sized for the largest of them, instead of one or two slabs per shape.  Sizes are requested rank-symmetrically (the plan
//    asks for the larger of the last / not-last device's count), so every rank finds the same entry.
//  * DFFT_IPC_POOL_MB (default: half of the device's memory) bounds what the pool holds: before a NEW buffer is allocated, idle
//    entries are evicted least-recently-used first until pool + request fit (entries in use are never touched -- the bound is on
//    the idle part).  Eviction is collective and deterministic (same pool, same order on every rank): peers unmap, rendezvous, the
//    owner frees.  The address range of an evicted buffer is remembered and never exported again (a fresh allocation that lands on
//    it is set aside and another one is taken): re-exporting recycled memory is exactly what the pool exists to avoid.
//  * Every reuse is a small rendezvous: all ranks must have picked the same registration and issued the same number of exchange
//    rounds on it (one all-reduce over the TCP rendezvous) -- which is also the hand-shake that every peer is done with the buffer's
//    previous plan before the new plan's plan-time input copy lands in it.
namespace {
int pool_evict(dfft_comm_t c, size_t idx) {
    const dfft_comm_s::PoolEntry e = c->pool[idx];
    trace("comm pool: evicting an idle entry", e.reg, (long long)e.bytes);
    (void)hipDeviceSynchronize();
    unsigned long long* own_flags = nullptr;
    for (int q = 0; q < c->P; ++q) {
        if (e.reg < (int)c->regs.size()) {
            void*& r = c->regs[e.reg][q];
            if (r && q != c->rank) (void)hipIpcCloseMemHandle(r);
            r = nullptr;
        }
        if (e.reg < (int)c->flag_regs.size()) {
            unsigned long long*& f = c->flag_regs[e.reg][q];
            if (f && q != c->rank) (void)hipIpcCloseMemHandle(f);
            if (q == c->rank) own_flags = f;
            f = nullptr;
        }
    }
    (void)hipGetLastError();
    const int rc = dfft_boot_barrier();  // every peer has unmapped: only now may the owners free
    if (own_flags) (void)hipFree(own_flags);
    (void)hipFree(e.buf);
    c->retired.emplace_back((uintptr_t)e.buf, e.bytes);
    c->pool.erase(c->pool.begin() + (long)idx);
    return rc;
}
bool pool_overlaps_retired(dfft_comm_t c, void* p, size_t bytes) {
    const uintptr_t a = (uintptr_t)p;
    for (const auto& r : c->retired)
        if (a < r.first + r.second && r.first < a + bytes) return true;
    return false;
}
std::string pool_role(const std::string& key) {
    const size_t cut = key.rfind(':');
    return cut == std::string::npos ? key : key.substr(cut + 1);
}
}  // namespace

int comm_recv_alloc(dfft_comm_t c, const std::string& key, size_t bytes, void** out) {
    *out = nullptr;
    if (!c || !c->pooled) {
        DFFT_HIP_TRY(hipMalloc(out, bytes ? bytes : 16));
        return DFFT_OK;
    }
    const std::string role = pool_role(key);
    int best = -1;
    for (int i = 0; i < (int)c->pool.size(); ++i) {
        const auto& e = c->pool[i];
        if (e.used || e.key != role || e.bytes < bytes) continue;
        if (best < 0 || e.bytes < c->pool[best].bytes) best = i;  // best fit; ties: the older registration
    }
    if (best >= 0) {
        auto& e = c->pool[best];
        // hand-shake: same registration, same number of rounds issued on it, on every rank
        const double sq = e.reg < (int)c->seq.size() ? (double)c->seq[e.reg] : 0.0;
        double       v[4] = {(double)e.reg, -(double)e.reg, sq, -sq};
        const int    rc = dfft_boot_allreduce_max(v, 4);
        if (rc) return rc;
        if (v[0] != -v[1] || v[2] != -v[3])
            return fail(DFFT_ECOMM, "comm_recv_alloc: the ranks of this communicator disagree about the pooled receive buffer a new plan takes over (registration " +
                                        std::to_string(e.reg) + ": plans must be created and destroyed in the same order on every rank)");
        e.used = true;
        e.stamp = ++c->pool_clock;
        *out = e.buf;
        trace("comm_recv_alloc: pooled buffer reused", e.reg, (long long)bytes);
        return DFFT_OK;
    }
    if (c->pool_budget > 0) {
        for (;;) {
            size_t held = 0;
            for (const auto& e : c->pool) held += e.bytes;
            if (held + bytes <= c->pool_budget) break;
            int lru = -1;
            for (int i = 0; i < (int)c->pool.size(); ++i)
                if (!c->pool[i].used && (lru < 0 || c->pool[i].stamp < c->pool[lru].stamp)) lru = i;
            if (lru < 0) break;  // everything the pool holds is in use: the bound is on the idle part
            const int rc = pool_evict(c, (size_t)lru);
            if (rc) return rc;
        }
    }
    std::vector<void*> aside;
    hipError_t         e = hipSuccess;
    for (int attempt = 0; attempt < 16; ++attempt) {
        e = hipMalloc(out, bytes ? bytes : 16);
        if (e != hipSuccess || !pool_overlaps_retired(c, *out, bytes ? bytes : 16)) break;
        aside.push_back(*out);  // lands on memory that was exported before: keep it out of the way and take another
        *out = nullptr;
    }
    for (void* a : aside) (void)hipFree(a);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(DFFT_EHIP, std::string("comm_recv_alloc: ") + hipGetErrorString(e));
    }
    if (!*out) return fail(DFFT_EHIP, "comm_recv_alloc: every fresh allocation landed on an address range this communicator has exported before");
    c->pending[*out] = role + "#" + std::to_string(bytes);
    return DFFT_OK;
}
// Is `buf` an allocation of comm_recv_alloc that no peer knows yet (not registered, not from the pool)?  Only such a buffer may still
// be exchanged for another one (placement of the receive buffer, dfft_plan.cpp); comm_recv_swap does it.
bool comm_recv_is_fresh(dfft_comm_t c, void* buf) {
    if (!c || !c->pooled) return true;
    return c->pending.find(buf) != c->pending.end();
}
int comm_recv_swap(dfft_comm_t c, void* old_buf, void* new_buf) {
    if (c && c->pooled) {
        auto it = c->pending.find(old_buf);
        if (it == c->pending.end()) return fail(DFFT_EINVAL, "comm_recv_swap: the buffer is already registered");
        const std::string key = it->second;
        c->pending.erase(it);
        c->pending[new_buf] = key;
    }
    DFFT_HIP_TRY(hipFree(old_buf));
    return DFFT_OK;
}
int comm_recv_free(dfft_comm_t c, void* buf) {
    if (!buf) return DFFT_OK;
    if (c && c->pooled) {
        for (auto& e : c->pool)
            if (e.buf == buf) {
                e.used = false;  // stays allocated, exported and mapped by the peers
                e.stamp = ++c->pool_clock;
                return DFFT_OK;
            }
        c->pending.erase(buf);
    }
    DFFT_HIP_TRY(hipFree(buf));
    return DFFT_OK;
}

// MAX of `flag` over all devices of the communicator -- how a failure that only ONE device can see (its one-launch YZ stage gave
// up, dfft_plan.cpp) becomes every device's return code instead of one rank's.  Host-synchronising and collective: called by every
// device from a host-synchronised execute, after its stream has drained.
int comm_agree_max(dfft_comm_t c, int me, int flag, hipStream_t stream, int* out) {
    *out = flag;
    if (!c || c->P <= 1) return DFFT_OK;
    trace("comm_agree_max enter", c->kind, flag);
    if (c->kind == 0) {
        {
            std::lock_guard<std::mutex> lk(c->m);
            if (flag > c->agree_in) c->agree_in = flag;
        }
        comm_thread_barrier(c);  // everyone has contributed
        {
            std::lock_guard<std::mutex> lk(c->m);
            *out = c->agree_in;
        }
        comm_thread_barrier(c);  // everyone has read
        if (me == 0) {
            std::lock_guard<std::mutex> lk(c->m);
            c->agree_in = 0;
        }
        return comm_thread_barrier(c);  // reset before anyone contributes to the next round
    }
    if (c->kind == 2 || c->kind == 3) {
        double v = (double)flag;
        const int rc = dfft_boot_allreduce_max(&v, 1);
        if (rc) return rc;
        *out = (int)v;
        return DFFT_OK;
    }
    // rccl: one-element all-reduce on the plan's stream
    if (!c->agree_dev) DFFT_HIP_TRY(hipMalloc((void**)&c->agree_dev, 2 * sizeof(int)));
    DFFT_HIP_TRY(hipMemcpyAsync(c->agree_dev, &flag, sizeof(int), hipMemcpyHostToDevice, stream));
    ncclResult_t r = ncclAllReduce(c->agree_dev, c->agree_dev + 1, 1, ncclInt, ncclMax, c->nccl, stream);
    if (r != ncclSuccess) return fail(DFFT_ERCCL, std::string("comm_agree_max: ncclAllReduce: ") + ncclGetErrorString(r));
    int got = flag;
    DFFT_HIP_TRY(hipMemcpyAsync(&got, c->agree_dev + 1, sizeof(int), hipMemcpyDeviceToHost, stream));
    DFFT_HIP_TRY(hipStreamSynchronize(stream));
    *out = got;
    return DFFT_OK;
}

int comm_register(dfft_comm_t c, int me, void* recvbuf, int device, int* reg_out) {
    if (me < 0 || me >= c->P || !reg_out) return fail(DFFT_EINVAL, "comm_register: index out of range");
    *reg_out = -1;
    if (c->kind == 0) {
        std::lock_guard<std::mutex> lk(c->m);
        const int reg = c->next_reg[me]++;
        if ((int)c->regs.size() <= reg) c->regs.resize(reg + 1, std::vector<void*>(c->P, nullptr));
        c->regs[reg][me] = recvbuf;
        c->devices[me] = device;
        *reg_out = reg;
    } else if (c->kind == 2 || c->kind == 3) {
        // collective over all processes: every rank publishes the IPC handle of its receive buffer and maps the others'
        if (me != c->rank) return fail(DFFT_EINVAL, "comm_register: plan index does not match the process rank");
        for (const auto& e : c->pool)
            if (e.buf == recvbuf) {  // a pooled buffer: registered (and mapped by every peer) since its first plan
                *reg_out = e.reg;
                return DFFT_OK;
            }
        const int reg = c->next_reg[me]++;
        trace("comm_register (ipc) enter", reg, c->kind);
        if ((int)c->regs.size() <= reg) c->regs.resize(reg + 1, std::vector<void*>(c->P, nullptr));
        auto share = [&](void* local, std::vector<void*>& out) -> int {
            // hipIpcGetMemHandle was seen failing with "invalid argument" for freshly recycled memory (round 5, tools/stall_hunt.py: the
            // second generation of plans of a process, four processes on one device, HSA_ENABLE_IPC_MODE_LEGACY=0).  Repeating the
            // call did not help in the two launches where it happened with the repeat in place -- what removed the failure is that
            // pooled communicators never export recycled memory (comm_recv_alloc) -- but a few spaced attempts cost nothing.  A rank
            // that fails here returns BEFORE the broadcasts below: its peers learn of it from the bounded rendezvous (connection
            // closed / time-out, DFFT_ECOMM with the reason) instead of waiting for a handle that never comes.
            hipIpcMemHandle_t mine;
            hipError_t        ge = hipErrorUnknown;
            for (int attempt = 0; attempt < 5 && ge != hipSuccess; ++attempt) {
                if (attempt) {
                    (void)hipGetLastError();
                    trace("hipIpcGetMemHandle failed, trying again", attempt, (long long)ge);
                    (void)hipDeviceSynchronize();
                    usleep(2000u << attempt);
                }
                ge = hipIpcGetMemHandle(&mine, local);
            }
            if (ge != hipSuccess)
                return fail(DFFT_EHIP, std::string("hipIpcGetMemHandle failed five times: ") + hipGetErrorString(ge) +
                                           " (is HSA_ENABLE_IPC_MODE_LEGACY=0 set, as this driver needs?)");
            for (int q = 0; q < c->P; ++q) {
                hipIpcMemHandle_t h = mine;
                int               rc = dfft_boot_bcast(&h, sizeof(h), q);
                if (rc) return rc;
                if (q == me) {
                    out[q] = local;
                } else {
                    void* ptr = nullptr;
                    DFFT_HIP_TRY(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
                    out[q] = ptr;
                }
            }
            return DFFT_OK;
        };
        int rc = share(recvbuf, c->regs[reg]);
        if (rc) return rc;
        for (int q = 0; q < c->P; ++q) c->devices[q] = -1;  // reached through IPC mappings, not through device ordinals
        if (c->kind == 3) {
            // flag words of this registration: fine-grained so that a peer's store is visible to a kernel spinning here
            if ((int)c->flag_regs.size() <= reg) {
                c->flag_regs.resize(reg + 1, std::vector<unsigned long long*>(c->P, nullptr));
                c->seq.resize(reg + 1, 0);
            }
            void* f = nullptr;
            DFFT_HIP_TRY(hipExtMallocWithFlags(&f, 2 * (size_t)c->P * sizeof(unsigned long long), hipDeviceMallocFinegrained));
            DFFT_HIP_TRY(hipMemset(f, 0, 2 * (size_t)c->P * sizeof(unsigned long long)));
            DFFT_HIP_TRY(hipDeviceSynchronize());
            std::vector<void*> fl(c->P, nullptr);
            rc = share(f, fl);
            if (rc) return rc;
            for (int q = 0; q < c->P; ++q) c->flag_regs[reg][q] = (unsigned long long*)fl[q];
            c->seq[reg] = 0;
        }
        *reg_out = reg;
        if (c->pooled) {
            auto it = c->pending.find(recvbuf);
            if (it != c->pending.end()) {
                const size_t      cut = it->second.rfind('#');
                dfft_comm_s::PoolEntry pe{it->second.substr(0, cut), recvbuf, (size_t)std::stoull(it->second.substr(cut + 1)), reg, true, ++c->pool_clock};
                c->pool.push_back(pe);
                c->pending.erase(it);
            }
        }
        trace("comm_register (ipc) done", reg, c->kind);
    } else {
        if (me != c->rank) return fail(DFFT_EINVAL, "comm_register: plan index does not match the RCCL rank");
        *reg_out = 0;  // receivers post their own ncclRecv: nothing to look up
    }
    return DFFT_OK;
}

int comm_unregister(dfft_comm_t c, int me, int reg) {
    if (reg < 0 || me < 0 || me >= c->P) return DFFT_OK;
    if (c->kind == 0) {
        std::lock_guard<std::mutex> lk(c->m);
        if (reg < (int)c->regs.size()) c->regs[reg][me] = nullptr;
    } else if ((c->kind == 2 || c->kind == 3) && reg < (int)c->regs.size()) {
        for (const auto& e : c->pool)
            if (e.reg == reg) return DFFT_OK;  // pooled: the registration outlives the plan (comm_recv_free hands it back)
        // collective (plans are destroyed in the same order everywhere): unmap the peers' buffers, and only then may their
        // owners free them
        trace("comm_unregister (ipc) enter", reg, c->kind);
        for (int q = 0; q < c->P; ++q) {
            void*& r = c->regs[reg][q];
            if (r && q != c->rank) (void)hipIpcCloseMemHandle(r);
            r = nullptr;
        }
        unsigned long long* own_flags = nullptr;
        if (c->kind == 3 && reg < (int)c->flag_regs.size()) {
            for (int q = 0; q < c->P; ++q) {
                unsigned long long*& f = c->flag_regs[reg][q];
                if (f && q != c->rank) (void)hipIpcCloseMemHandle(f);
                if (q == c->rank) own_flags = f;
                f = nullptr;
            }
        }
        (void)hipGetLastError();
        int rc = dfft_boot_barrier();
        if (own_flags) (void)hipFree(own_flags);
        return rc;
    }
    return DFFT_OK;
}

namespace {
// One message of a round: element offsets/counts on the send side, the receive side, and (push-style local exchange) the
// offset inside the peer's receive buffer.
struct Msg {
    int       peer;
    long long so, sc;  // send to `peer`
    long long ro, rc;  // receive from `peer`
    long long doff;    // where the sent block lands in `peer`'s receive buffer
};
using Round = std::vector<Msg>;

Round whole_round(const ExchangeDesc& x) {
    Round r;
    for (int q = 0; q < x.P; ++q) r.push_back(Msg{q, x.soffset[q], x.scount[q], x.roffset[q], x.rcount[q], x.doffset[q]});
    return r;
}

// Messages of X-plane part k (and Y sub-block `ycut`, or all of them for -1), ordered by peer then sub-block: both ends of
// a pair enumerate their common messages in the same order, which is what RCCL's send/recv matching needs.
// Backward exchange (Y slabs -> X slabs), the mirror image: what a forward piece receives is what the backward piece
// sends, and vice versa.  Send buffer [k][x all][y in k][N2] (the inverse X pass of Y sub-block k writes it), receive
// buffer [k][src][x mine][y in k][N2] (ycuts = 1: [x all][y mine][N2] and the reference's packed [src][x][y_src][N2]).
Round part_round_backward(const ExchangeDesc& x, int k, long long cp, int ycut) {
    Round r;
    long long mx0, mnx;
    part_range(x.xsize[x.me], cp, k, &mx0, &mnx);
    std::vector<long long> xstart(x.P + 1, 0);
    for (int q = 0; q < x.P; ++q) xstart[q + 1] = xstart[q] + x.xsize[q];
    if (x.ycuts <= 1) {
        for (int q = 0; q < x.P; ++q) {
            long long qx0, qnx;
            part_range(x.xsize[q], cp, k, &qx0, &qnx);
            Msg m;
            m.peer = q;
            // q's planes [qx0, qx0+qnx) out of my [x all][y mine][N2]
            m.so = x.soffset[q] + qx0 * x.ysize[x.me] * x.n2;
            m.sc = qnx * x.ysize[x.me] * x.n2;
            // where they land at q: block `me` of its packed [src][x of q][y_src][N2]
            m.doff = x.doffset[q] + qx0 * x.ysize[x.me] * x.n2;
            // my planes [mx0, mx0+mnx) from q: block q of my packed buffer
            m.ro = x.roffset[q] + mx0 * x.ysize[q] * x.n2;
            m.rc = mnx * x.ysize[q] * x.n2;
            r.push_back(m);
        }
        return r;
    }
    const int       K = x.ycuts;
    const long long ysub = x.ysize[0] / K, row = ysub * x.n2, n0 = xstart[x.P];
    for (int q = 0; q < x.P; ++q) {
        long long qx0, qnx;
        part_range(x.xsize[q], cp, k, &qx0, &qnx);
        for (int y = 0; y < K; ++y) {
            if (ycut >= 0 && y != ycut) continue;
            Msg m;
            m.peer = q;
            m.so = (long long)y * n0 * row + (xstart[q] + qx0) * row;                  // [y][x global][ysub][N2]
            m.sc = qnx * row;
            m.doff = ((long long)y * x.P + x.me) * x.xsize[q] * row + qx0 * row;       // [y][src][x of q][ysub][N2] at q
            m.ro = ((long long)y * x.P + q) * x.xsize[x.me] * row + mx0 * row;
            m.rc = mnx * row;
            r.push_back(m);
        }
    }
    return r;
}

Round part_round(const ExchangeDesc& x, int k, long long cp, int ycut) {
    if (x.direction == DFFT_BACKWARD) return part_round_backward(x, k, cp, ycut);
    Round r;
    long long mx0, mnx;
    part_range(x.xsize[x.me], cp, k, &mx0, &mnx);
    if (x.ycuts <= 1) {
        for (int q = 0; q < x.P; ++q) {
            Msg m;
            m.peer = q;
            // my planes [mx0, mx0+mnx) of chunk(me -> q)
            m.so = x.soffset[q] + mx0 * x.ysize[q] * x.n2;
            m.sc = mnx * x.ysize[q] * x.n2;
            m.doff = x.doffset[q] + mx0 * x.ysize[q] * x.n2;
            // q's planes [qx0, qx0+qnx) of chunk(q -> me)
            long long qx0, qnx;
            part_range(x.xsize[q], cp, k, &qx0, &qnx);
            m.ro = x.roffset[q] + qx0 * x.ysize[x.me] * x.n2;
            m.rc = qnx * x.ysize[x.me] * x.n2;
            r.push_back(m);
        }
        return r;
    }
    const int       K = x.ycuts;
    const long long ysub = x.ysize[0] / K;  // uniform by construction (fill_exchange)
    std::vector<long long> xstart(x.P + 1, 0);
    for (int q = 0; q < x.P; ++q) xstart[q + 1] = xstart[q] + x.xsize[q];
    const long long n0 = xstart[x.P], row = ysub * x.n2;
    for (int q = 0; q < x.P; ++q) {
        long long qx0, qnx;
        part_range(x.xsize[q], cp, k, &qx0, &qnx);
        for (int y = 0; y < K; ++y) {
            if (ycut >= 0 && y != ycut) continue;
            Msg m;
            m.peer = q;
            m.so = ((long long)y * x.P + q) * x.xsize[x.me] * row + mx0 * row;  // [y][dst][x][ysub][N2] (even X split)
            m.sc = mnx * row;
            m.doff = (long long)y * n0 * row + (xstart[x.me] + mx0) * row;     // [y][x global][ysub][N2] at the peer
            m.ro = (long long)y * n0 * row + (xstart[q] + qx0) * row;
            m.rc = qnx * row;
            r.push_back(m);
        }
    }
    return r;
}

int exchange_local(dfft_comm_t c, const ExchangeDesc& x, const Round& r, hipStream_t stream) {
    const size_t eb = elem_bytes(x.dtype);
    trace("exchange (host-synchronised) enter", x.slot, (long long)r.size());
    // every device has finished producing its send buffer and consuming its receive buffer
    DFFT_HIP_TRY(hipStreamSynchronize(stream));
    {
        const int brc = comm_thread_barrier(c);  // IPC: the TCP rendezvous -- a dead peer must not let this rank push on
        if (brc) return brc;
    }
    int mydev = 0;
    DFFT_HIP_TRY(hipGetDevice(&mydev));
    const size_t nm = r.size();
    const size_t first = nm ? (nm * (size_t)x.me) / (size_t)x.P : 0;  // stagger the targets like a rotation schedule
    for (size_t i = 0; i < nm; ++i) {
        const Msg& m = r[(first + i) % nm];
        if (m.sc == 0) continue;
        void* dstbase;
        int   dstdev;
        {
            std::lock_guard<std::mutex> lk(c->m);
            dstbase = (x.slot >= 0 && x.slot < (int)c->regs.size()) ? c->regs[x.slot][m.peer] : nullptr;
            dstdev = c->devices[m.peer];
        }
        if (!dstbase) return fail(DFFT_ECOMM, "local exchange: peer plan is not registered");
        // chunk(me -> peer) lands in peer's bufferDev1 at the offset reserved there for source `me`
        // (recv_offset, fft_mpi_3d_api.cpp:618-625)
        char*        dst = (char*)dstbase + (size_t)m.doff * eb;
        const char*  src = (const char*)x.sendbuf + (size_t)m.so * eb;
        const size_t bytes = (size_t)m.sc * eb;
        if (dstdev == mydev || dstdev < 0) DFFT_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream));
        else DFFT_HIP_TRY(hipMemcpyPeerAsync(dst, dstdev, src, mydev, bytes, stream));
    }
    DFFT_HIP_TRY(hipStreamSynchronize(stream));
    return comm_thread_barrier(c);  // all incoming chunks have landed
}

// Asynchronous IPC exchange (kind 3): everything is enqueued on `stream`, nothing blocks the host.
//   sync 1: tell every source "my receive regions of this round are free", wait for the same word from every destination
//   push  : device-to-device copies into the peers' IPC-mapped receive buffers (SDMA engines, no CUs)
//   sync 2: tell every destination "your data has landed", wait for the same word from every source
// Both ends issue the rounds of a registration in the same order, so one monotonically increasing word per pair suffices.
int exchange_ipc_async(dfft_comm_t c, const ExchangeDesc& x, const Round& r, hipStream_t stream) {
    const size_t eb = elem_bytes(x.dtype);
    if (x.slot < 0 || x.slot >= (int)c->flag_regs.size()) return fail(DFFT_ECOMM, "ipc exchange: plan is not registered");
    const unsigned long long seq = ++c->seq[x.slot];
    const int                P = c->P, me = c->rank;
    trace("exchange (ipc, stream-ordered) queue round", x.slot, (long long)seq);
    std::vector<char>        is_src(P, 0), is_dst(P, 0);
    for (const Msg& m : r) {
        if (m.peer == me) continue;
        if (m.rc > 0) is_src[m.peer] = 1;
        if (m.sc > 0) is_dst[m.peer] = 1;
    }
    unsigned long long* mine = c->flag_regs[x.slot][me];
    IpcSyncLists        ready, arrive;
    ready.nsig = ready.nwait = arrive.nsig = arrive.nwait = 0;
    for (int q = 0; q < P; ++q) {
        if (is_src[q]) {  // q will push into me
            ready.sig[ready.nsig++] = c->flag_regs[x.slot][q] + me;         // q's ready[me]
            arrive.wait[arrive.nwait++] = mine + P + q;                      // my arrive[q]
        }
        if (is_dst[q]) {  // I will push into q
            ready.wait[ready.nwait++] = mine + q;                            // my ready[q]
            arrive.sig[arrive.nsig++] = c->flag_regs[x.slot][q] + P + me;   // q's arrive[me]
        }
    }
    (void)hipGetLastError();
    if (ready.nsig || ready.nwait) hipLaunchKernelGGL(ipc_sync_kernel, dim3(1), dim3(64), 0, stream, ready, seq, c->err, ipc_timeout_ticks());
    // every peer has its own xGMI link: the pushes to different peers go out on per-peer helper streams (fork after sync 1,
    // join before sync 2) so that the copy engines drive all links at once instead of one after the other
    if (c->peer_streams.empty()) {
        c->peer_streams.assign(P, nullptr);
        c->peer_events.assign(P, nullptr);
        DFFT_HIP_TRY(hipEventCreateWithFlags(&c->fork_event, hipEventDisableTiming));
        for (int q = 0; q < P; ++q) {
            if (q == me) continue;
            DFFT_HIP_TRY(hipStreamCreateWithFlags(&c->peer_streams[q], hipStreamNonBlocking));
            DFFT_HIP_TRY(hipEventCreateWithFlags(&c->peer_events[q], hipEventDisableTiming));
        }
    }
    DFFT_HIP_TRY(hipEventRecord(c->fork_event, stream));
    std::vector<char> used(P, 0);
    for (const Msg& m : r) {
        if (m.sc == 0) continue;
        void* dstbase = c->regs[x.slot][m.peer];
        if (!dstbase) return fail(DFFT_ECOMM, "ipc exchange: peer buffer is not mapped");
        hipStream_t s = stream;
        if (m.peer != me) {
            s = c->peer_streams[m.peer];
            if (!used[m.peer]) DFFT_HIP_TRY(hipStreamWaitEvent(s, c->fork_event, 0));
            used[m.peer] = 1;
        }
        DFFT_HIP_TRY(hipMemcpyAsync((char*)dstbase + (size_t)m.doff * eb, (const char*)x.sendbuf + (size_t)m.so * eb,
                                    (size_t)m.sc * eb, hipMemcpyDeviceToDevice, s));
    }
    for (int q = 0; q < P; ++q) {
        if (!used[q]) continue;
        DFFT_HIP_TRY(hipEventRecord(c->peer_events[q], c->peer_streams[q]));
        DFFT_HIP_TRY(hipStreamWaitEvent(stream, c->peer_events[q], 0));
    }
    if (arrive.nsig || arrive.nwait) hipLaunchKernelGGL(ipc_sync_kernel, dim3(1), dim3(64), 0, stream, arrive, seq, c->err, ipc_timeout_ticks());
    DFFT_HIP_TRY(hipGetLastError());
    return DFFT_OK;
}

int exchange_rccl(dfft_comm_t c, const ExchangeDesc& x, const Round& r, hipStream_t stream) {
    const size_t         eb = elem_bytes(x.dtype);
    const ncclDataType_t ty = x.dtype == DFFT_F64 ? ncclDouble : ncclFloat;
    // self chunk: plain device copy (no link involved).  DFFT_RCCL_SELF_SENDRECV=1 sends it through ncclSend/ncclRecv
    // instead, so that the grouped send/recv code below can be exercised with a single GPU (tests).
    static const bool self_via_rccl = [] {
        const char* e = std::getenv("DFFT_RCCL_SELF_SENDRECV");
        return e && *e && *e != '0';
    }();
    bool remote = false;
    for (const Msg& m : r) {
        if (m.peer != x.me || self_via_rccl) {
            remote = remote || m.sc > 0 || m.rc > 0;
            continue;
        }
        if (m.sc > 0)
            DFFT_HIP_TRY(hipMemcpyAsync((char*)x.recvbuf + (size_t)m.ro * eb, (const char*)x.sendbuf + (size_t)m.so * eb,
                                        (size_t)m.sc * eb, hipMemcpyDeviceToDevice, stream));
    }
    if (!remote) return DFFT_OK;
    trace("exchange (rccl) group start", x.me, (long long)r.size());
    ncclResult_t rc = ncclGroupStart();
    if (rc != ncclSuccess) return fail(DFFT_ERCCL, std::string("ncclGroupStart: ") + ncclGetErrorString(rc));
    // rotation schedule: at distance i send to me+i and receive from me-i (messages of one peer keep their order)
    for (int i = self_via_rccl ? 0 : 1; i < x.P; ++i) {
        const int to = (x.me + i) % x.P, from = (x.me - i + x.P) % x.P;
        for (const Msg& m : r) {
            if (m.peer == to && m.sc > 0) {
                rc = ncclSend((const char*)x.sendbuf + (size_t)m.so * eb, (size_t)m.sc * 2, ty, to, c->nccl, stream);
                if (rc != ncclSuccess) {
                    (void)ncclGroupEnd();  // never leave the group open behind an error
                    return fail(DFFT_ERCCL, std::string("ncclSend: ") + ncclGetErrorString(rc));
                }
            }
        }
        for (const Msg& m : r) {
            if (m.peer == from && m.rc > 0) {
                rc = ncclRecv((char*)x.recvbuf + (size_t)m.ro * eb, (size_t)m.rc * 2, ty, from, c->nccl, stream);
                if (rc != ncclSuccess) {
                    (void)ncclGroupEnd();
                    return fail(DFFT_ERCCL, std::string("ncclRecv: ") + ncclGetErrorString(rc));
                }
            }
        }
    }
    rc = ncclGroupEnd();
    trace("exchange (rccl) group end", x.me, (long long)rc);
    if (rc != ncclSuccess) return fail(DFFT_ERCCL, std::string("ncclGroupEnd: ") + ncclGetErrorString(rc));
    return DFFT_OK;
}
}  // namespace

void comm_part_messages(const ExchangeDesc& x, int k, long long cp, int ycut, std::vector<int>& peer, std::vector<long long>& so,
                        std::vector<long long>& sc, std::vector<long long>& ro, std::vector<long long>& rc) {
    for (const Msg& m : part_round(x, k, cp, ycut)) {
        peer.push_back(m.peer);
        so.push_back(m.so);
        sc.push_back(m.sc);
        ro.push_back(m.ro);
        rc.push_back(m.rc);
    }
}

// DFFT_EXCHANGE_NOOP=1 (measurement hook, tools/local_by_P.py): the exchange moves nothing and synchronises nobody, so that ONE
// rank of a P-rank plan can be executed alone and its local stages (t0, t3) timed with the real P > 1 address maps.  The
// results are garbage by construction.
static bool exchange_noop() {
    static const bool on = [] {
        const char* e = getenv("DFFT_EXCHANGE_NOOP");
        return e && *e && *e != '0';
    }();
    return on;
}

int comm_exchange(dfft_comm_t c, const ExchangeDesc& x, hipStream_t stream) {
    if (exchange_noop()) return DFFT_OK;
    const Round r = whole_round(x);
    if (c->kind == 3) return exchange_ipc_async(c, x, r, stream);
    if (c->kind != 1) return exchange_local(c, x, r, stream);
    return exchange_rccl(c, x, r, stream);
}

int comm_exchange_part(dfft_comm_t c, const ExchangeDesc& x, int k, long long cp, hipStream_t stream, int ycut) {
    if (exchange_noop()) return DFFT_OK;
    if ((int)x.xsize.size() != x.P || (int)x.ysize.size() != x.P || cp < 1 || k < 0 || ycut >= x.ycuts)
        return fail(DFFT_EINVAL, "comm_exchange_part: descriptor has no plane geometry");
    const Round r = part_round(x, k, cp, ycut);
    if (c->kind == 3) return exchange_ipc_async(c, x, r, stream);
    if (c->kind != 1) return exchange_local(c, x, r, stream);
    return exchange_rccl(c, x, r, stream);
}

}  // namespace dfft

using namespace dfft;

extern "C" {

int dfft_comm_create_local(int total_devices, dfft_comm_t* comm) {
    if (total_devices < 1 || !comm) return fail(DFFT_EINVAL, "dfft_comm_create_local: bad arguments");
    dfft_comm_s* c = new dfft_comm_s;
    c->kind = 0;
    c->P = total_devices;
    c->next_reg.assign(total_devices, 0);
    c->devices.assign(total_devices, 0);
    *comm = c;
    return DFFT_OK;
}

int dfft_comm_create_ipc(int total_devices, int global_idx, int async_exchange, dfft_comm_t* comm) {
    if (total_devices < 1 || !comm || global_idx < 0 || global_idx >= total_devices || total_devices > 64)
        return fail(DFFT_EINVAL, "dfft_comm_create_ipc: bad arguments (at most 64 devices)");
    int rc = dfft_boot_init();
    if (rc) return rc;
    if (dfft_boot_size() != total_devices || dfft_boot_rank() != global_idx)
        return fail(DFFT_ECOMM, "dfft_comm_create_ipc: one process per device -- the rendezvous' rank/size must equal global_idx/total_devices");
    dfft_comm_s* c = new dfft_comm_s;
    c->kind = async_exchange ? 3 : 2;
    c->P = total_devices;
    c->rank = global_idx;
    {
        const char* pe = getenv("DFFT_IPC_POOL");
        c->pooled = !(pe && *pe == '0');
    }
    c->next_reg.assign(total_devices, 0);
    c->devices.assign(total_devices, -1);
    if (hipGetDevice(&c->device) != hipSuccess) {
        delete c;
        return fail(DFFT_ENOGPU, "dfft_comm_create_ipc: no HIP device");
    }
    {
        // what the pool of receive buffers may hold before idle entries are evicted: DFFT_IPC_POOL_MB, default half of the device's memory
        const char* mb = getenv("DFFT_IPC_POOL_MB");
        size_t      free_b = 0, total_b = 0;
        if (mb && atoll(mb) >= 0) c->pool_budget = (size_t)atoll(mb) << 20;
        else if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) c->pool_budget = total_b / 2;
        else (void)hipGetLastError();
    }
    if (c->kind == 3) {
        void* e = nullptr;
        if (hipHostMalloc(&e, sizeof(unsigned long long), hipHostMallocMapped) != hipSuccess) {
            delete c;
            return fail(DFFT_EHIP, "dfft_comm_create_ipc: hipHostMalloc");
        }
        c->err = (unsigned long long*)e;
        *c->err = 0;
    }
    *comm = c;
    return DFFT_OK;
}

int dfft_rccl_unique_id(char id[128]) {
    static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id is 128 bytes");
    ncclUniqueId u;
    ncclResult_t r = ncclGetUniqueId(&u);
    if (r != ncclSuccess) return fail(DFFT_ERCCL, std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
    std::memcpy(id, &u, 128);
    return DFFT_OK;
}

int dfft_comm_create_rccl(const char id[128], int total_devices, int global_idx, dfft_comm_t* comm) {
    if (!id || !comm || total_devices < 1 || global_idx < 0 || global_idx >= total_devices)
        return fail(DFFT_EINVAL, "dfft_comm_create_rccl: bad arguments");
    ncclUniqueId u;
    std::memcpy(&u, id, 128);
    dfft_comm_s* c = new dfft_comm_s;
    c->kind = 1;
    c->P = total_devices;
    c->rank = global_idx;
    if (hipGetDevice(&c->device) != hipSuccess) {
        delete c;
        return fail(DFFT_ENOGPU, "dfft_comm_create_rccl: no HIP device");
    }
    trace("ncclCommInitRank enter", global_idx, total_devices);
    ncclResult_t r = ncclCommInitRank(&c->nccl, total_devices, u, global_idx);
    trace("ncclCommInitRank returned", global_idx, (long long)r);
    if (r != ncclSuccess) {
        delete c;
        return fail(DFFT_ERCCL, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
    }
    *comm = c;
    return DFFT_OK;
}

int dfft_comm_info(dfft_comm_t comm, int* kind, int* size, int* rank, int* device) {
    if (!comm) return fail(DFFT_EINVAL, "dfft_comm_info: null communicator");
    int sz = comm->P, rk = comm->rank, dev = comm->kind == 0 ? -1 : comm->device;
    if (comm->kind == 1) {
        ncclResult_t r = ncclCommCount(comm->nccl, &sz);
        if (r == ncclSuccess) r = ncclCommUserRank(comm->nccl, &rk);
        if (r == ncclSuccess) r = ncclCommCuDevice(comm->nccl, &dev);
        if (r != ncclSuccess) return fail(DFFT_ERCCL, std::string("dfft_comm_info: ") + ncclGetErrorString(r));
    }
    if (kind) *kind = comm->kind;
    if (size) *size = sz;
    if (rank) *rank = rk;
    if (device) *device = dev;
    return DFFT_OK;
}

int dfft_comm_destroy(dfft_comm_t comm) {
    if (!comm) return DFFT_OK;
    trace("dfft_comm_destroy", comm->kind, comm->rank);
    if (comm->kind == 1 && comm->nccl) ncclCommDestroy(comm->nccl);
    if (!comm->pool.empty()) {
        // pooled registrations: unmap the peers' buffers and flag words, meet the peers (best effort: a rank that has died must not
        // keep the others from leaving), then free this rank's own
        (void)hipDeviceSynchronize();
        for (const auto& e : comm->pool) {
            for (int q = 0; q < comm->P; ++q) {
                if (q == comm->rank) continue;
                if (e.reg < (int)comm->regs.size() && comm->regs[e.reg][q]) (void)hipIpcCloseMemHandle(comm->regs[e.reg][q]);
                if (e.reg < (int)comm->flag_regs.size() && comm->flag_regs[e.reg][q]) (void)hipIpcCloseMemHandle(comm->flag_regs[e.reg][q]);
            }
        }
        (void)hipGetLastError();
        (void)dfft_boot_barrier();
        for (const auto& e : comm->pool) {
            if (e.reg < (int)comm->flag_regs.size() && comm->flag_regs[e.reg][comm->rank]) (void)hipFree(comm->flag_regs[e.reg][comm->rank]);
            (void)hipFree(e.buf);
        }
        comm->pool.clear();
    }
    if (comm->err) (void)hipHostFree(comm->err);
    if (comm->agree_dev) (void)hipFree(comm->agree_dev);
    for (hipStream_t s : comm->peer_streams)
        if (s) {
            (void)hipStreamSynchronize(s);
            (void)hipStreamDestroy(s);
        }
    for (hipEvent_t e : comm->peer_events)
        if (e) (void)hipEventDestroy(e);
    if (comm->fork_event) (void)hipEventDestroy(comm->fork_event);
    delete comm;
    return DFFT_OK;
}

}  // extern "C"
