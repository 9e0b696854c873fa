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

// Device twiddle table e^{-2 pi i k / n}, k < n, cached per (device, n, dtype).
int get_twiddles(int n, int dtype, const void** table);

// ---- exchange ------------------------------------------------------------------------------------------------------
struct ExchangeDesc {
    int                    dtype;
    int                    P, me;
    void*                  sendbuf;  // bufferDev2
    void*                  recvbuf;  // bufferDev1 (this device's; peers' are looked up through the communicator)
    std::vector<long long> scount, soffset, rcount, roffset;  // elements, indexed by peer
    std::vector<long long> doffset;  // where chunk(me -> peer) lands inside peer's recvbuf (push-style local exchange)
};

int comm_register(dfft_comm_t comm, int me, void* recvbuf, int device);
int comm_unregister(dfft_comm_t comm, int me);
int comm_kind(dfft_comm_t comm);  // 0 local, 1 rccl
int comm_size(dfft_comm_t comm);
// Local: host-synchronising collective (thread barrier + peer copies); RCCL: enqueued on `stream`.
int comm_exchange(dfft_comm_t comm, const ExchangeDesc& x, hipStream_t stream);
// Thread barrier over the P local device-threads (no-op for RCCL communicators).
int comm_thread_barrier(dfft_comm_t comm);

}  // namespace dfft
