// dfft_internal.h -- shared host-side declarations of libdfft_mi355x (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>

#include "../../include/dfft.h"
#include "dfft_kernels.h"

namespace dfft {

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
// control-plane event ring (dfft_trace.cpp): `what` must be a string literal; printed on communication errors, on SIGUSR2
// (DFFT_TRACE_SIGNAL=1) and by dfft_trace_dump()
void trace(const char* what, long long a = 0, long long b = 0);
void trace_on_error(int code, const std::string& msg);

#define DFFT_HIP_TRY(stmt)                                                                                          \
    do {                                                                                                            \
        hipError_t e_ = (stmt);                                                                                     \
        if (e_ != hipSuccess)                                                                                       \
            return ::dfft::fail(DFFT_EHIP, std::string(#stmt) + " failed: " + hipGetErrorString(e_) + " [" +        \
                                               __FILE__ + ":" + std::to_string(__LINE__) + "]");                    \
    } while (0)

inline size_t elem_bytes(int dtype) { return dtype == DFFT_F64 ? 16 : 8; }

// Slab decomposition of one axis: ceil split, the last device takes the remainder
// (fft_mpi_3d_api.cpp:84-91).
struct Slab {
    long long n;   // axis length
    int       P;   // devices
    long long blk; // ceil(n / P)
    long long size(int g) const { return g < P - 1 ? blk : n - (long long)(P - 1) * blk; }
    long long start(int g) const { return (long long)g * blk; }
};
inline Slab make_slab(long long n, int P) { return Slab{n, P, (n + P - 1) / P}; }

// Plan-owned slabs (hand-over buffer): hipMalloc, or one virtual range backed by physical chunks of a chosen size
// (dfft_alloc.cpp; DFFT_W_ALLOC).  slab_free accepts pointers of either kind.
hipError_t slab_alloc(void** p, size_t bytes);
hipError_t slab_free(void* p);

// Device twiddle table e^{-2 pi i k / n}, k < n, cached per (device, n, dtype).
int get_twiddles(int n, int dtype, const void** table);

// ---- exchange ------------------------------------------------------------------------------------------------------
struct ExchangeDesc {
    int                    dtype;
    int                    P, me;
    int                    direction = DFFT_FORWARD;  // X -> Y slabs (forward) or Y -> X slabs (backward)
    int                    slot = -1;  // registration id (comm_register): which receive buffer of the peers to push into
    void*                  sendbuf;  // bufferDev2
    void*                  recvbuf;  // bufferDev1 (this device's; peers' are looked up through the communicator)
    std::vector<long long> scount, soffset, rcount, roffset;  // elements, indexed by peer
    std::vector<long long> doffset;  // where chunk(me -> peer) lands inside peer's recvbuf (push-style local exchange)
    // forward exchange only: what is needed to cut every message along X into plane chunks (t0/t2 overlap).
    // chunk(src -> dst) is [x in src][y in dst][N2] with x slowest, so planes [x0, x0+nx) of it are contiguous on both
    // sides: + x0 * ysize[dst] * n2 elements.
    std::vector<long long> xsize, ysize;  // planes owned by each device before / rows owned after the transform
    long long              n2 = 0;
    // t2/t3 overlap: every destination's Y range is additionally cut into `ycuts` equal sub-blocks (only when all ysize
    // and all xsize are equal and divisible).  Send layout [k][dst][x][y in sub-block k][N2], receive layout
    // [k][x (all sources)][y][N2]: sub-block k of the receive buffer is a complete [N0][ysize/ycuts][N2] slab the X pass
    // can start on, and sub-block k of the send buffer is the region that X pass overwrites with its result.
    int                    ycuts = 1;
};

// Range of planes device g contributes to exchange part k when every device cuts its slab into parts of `cp` planes.
inline void part_range(long long xs_g, long long cp, int k, long long* x0, long long* nx) {
    long long a = (long long)k * cp, b = a + cp;
    if (a > xs_g) a = xs_g;
    if (b > xs_g) b = xs_g;
    *x0 = a;
    *nx = b - a;
}

// Registers device `me`'s receive buffer and returns its registration id (ExchangeDesc::slot).  Plans must be created in
// the same order on every device thread / process; the IPC communicator makes this call collective.
int comm_register(dfft_comm_t comm, int me, void* recvbuf, int device, int* reg);
int comm_unregister(dfft_comm_t comm, int me, int reg);
// Device memory for a buffer that a plan is going to register as a receive buffer (`key`: everything that determines its size
// and role, identical on every rank), and its release.  IPC communicators keep such buffers and their registrations for their
// whole life and reuse them (dfft_exchange.cpp, comm_recv_alloc); everywhere else these are hipMalloc / hipFree.
int comm_recv_alloc(dfft_comm_t comm, const std::string& key, size_t bytes, void** out);
int comm_recv_free(dfft_comm_t comm, void* buf);
bool comm_recv_is_fresh(dfft_comm_t comm, void* buf);              // not registered with / known to any peer yet
int comm_recv_swap(dfft_comm_t comm, void* old_buf, void* new_buf);  // frees old_buf, new_buf takes its place (fresh buffers only)
int comm_kind(dfft_comm_t comm);  // 0 local, 1 rccl, 2 ipc (host-synchronised), 3 ipc (stream-ordered)
bool comm_is_async(dfft_comm_t comm);  // exchanges are enqueued on the stream (rccl, ipc-async) instead of blocking the host
int comm_check(dfft_comm_t comm);      // error reported by an asynchronous exchange since the last check
int comm_size(dfft_comm_t comm);
// Local: host-synchronising collective (thread barrier + peer copies); RCCL: enqueued on `stream`.
int comm_exchange(dfft_comm_t comm, const ExchangeDesc& x, hipStream_t stream);
// Forward exchange restricted to part k (planes [k*cp, (k+1)*cp) of every source slab); the parts of k = 0..K-1 together
// move exactly what comm_exchange moves.  ycut >= 0 further restricts it to Y sub-block `ycut` of every destination
// (x.ycuts > 1), -1 moves all sub-blocks.  RCCL: enqueued on `stream`; LOCAL: host-synchronising like comm_exchange.
int comm_exchange_part(dfft_comm_t comm, const ExchangeDesc& x, int k, long long cp, hipStream_t stream, int ycut = -1);
// The messages comm_exchange_part issues (same order), for inspection through the C-ABI (dfft_exchange_part_layout).
void comm_part_messages(const ExchangeDesc& x, int k, long long cp, int ycut, std::vector<int>& peer, std::vector<long long>& so,
                        std::vector<long long>& sc, std::vector<long long>& ro, std::vector<long long>& rc);
// MAX of `flag` over all devices (host-synchronising, collective): turns a failure only one device can see into every device's.
int comm_agree_max(dfft_comm_t comm, int me, int flag, hipStream_t stream, int* out);
// Thread barrier over the P local device-threads (no-op for RCCL communicators).
int comm_thread_barrier(dfft_comm_t comm);

}  // namespace dfft
