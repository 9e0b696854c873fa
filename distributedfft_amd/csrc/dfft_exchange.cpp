// dfft_exchange.cpp -- t2, the personalised all-to-all that re-slabs X -> Y.
//
// Replaces slabAlltoall (fft_mpi_3d_api.cpp:610-672): the reference moves chunk (src -> dst) with hipMemcpyPeerAsync
// inside a process and with MPI_Isend/Irecv on device pointers (UCX) between processes.  Here:
//   RCCL  communicator : one grouped ncclSend/ncclRecv per peer on the plan's stream (xGMI point-to-point: each
//                        (src,dst) pair has its own link, so the whole exchange is bounded by the pair chunk
//                        S*N/P^2 over ~153 GB/s), self chunk by device copy.
//   LOCAL communicator : P device-threads of one process; peer copies between the plans' buffers with the same
//                        barrier discipline as the reference's `#pragma omp barrier`s (fft_mpi_3d_api.cpp:190,195).
//                        Also used with P virtual devices on one physical GPU for single-GPU parity tests.
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <mutex>

#include "dfft_internal.h"

struct dfft_comm_s {
    int kind = 0;  // 0 local, 1 rccl
    int P = 1;
    // local
    std::mutex              m;
    std::condition_variable cv;
    int                     arrived = 0;
    unsigned long           generation = 0;
    std::vector<void*>      recvbufs;
    std::vector<int>        devices;
    // rccl
    ncclComm_t nccl = nullptr;
    int        rank = 0;
    int        device = 0;
};

namespace dfft {

int comm_kind(dfft_comm_t c) { return c->kind; }
int comm_size(dfft_comm_t c) { return c->P; }

int comm_thread_barrier(dfft_comm_t c) {
    if (c->kind != 0 || c->P <= 1) return DFFT_OK;
    std::unique_lock<std::mutex> lk(c->m);
    const unsigned long gen = c->generation;
    if (++c->arrived == c->P) {
        c->arrived = 0;
        ++c->generation;
        c->cv.notify_all();
    } else {
        c->cv.wait(lk, [&] { return c->generation != gen; });
    }
    return DFFT_OK;
}

int comm_register(dfft_comm_t c, int me, void* recvbuf, int device) {
    if (me < 0 || me >= c->P) return fail(DFFT_EINVAL, "comm_register: device index out of range");
    if (c->kind == 0) {
        std::lock_guard<std::mutex> lk(c->m);
        c->recvbufs[me] = recvbuf;
        c->devices[me] = device;
    } else {
        if (me != c->rank) return fail(DFFT_EINVAL, "comm_register: plan index does not match the RCCL rank");
    }
    return DFFT_OK;
}

int comm_unregister(dfft_comm_t c, int me) {
    if (c->kind == 0 && me >= 0 && me < c->P) {
        std::lock_guard<std::mutex> lk(c->m);
        c->recvbufs[me] = nullptr;
    }
    return DFFT_OK;
}

static int exchange_local(dfft_comm_t c, const ExchangeDesc& x, hipStream_t stream) {
    const size_t eb = elem_bytes(x.dtype);
    // every device has finished producing its send buffer and consuming its receive buffer
    DFFT_HIP_TRY(hipStreamSynchronize(stream));
    comm_thread_barrier(c);
    int mydev = 0;
    DFFT_HIP_TRY(hipGetDevice(&mydev));
    for (int i = 0; i < x.P; ++i) {
        const int peer = (x.me + i) % x.P;  // stagger the targets like a rotation schedule
        if (x.scount[peer] == 0) continue;
        void* dstbase;
        int   dstdev;
        {
            std::lock_guard<std::mutex> lk(c->m);
            dstbase = c->recvbufs[peer];
            dstdev = c->devices[peer];
        }
        if (!dstbase) return fail(DFFT_ECOMM, "local exchange: peer plan is not registered");
        // chunk(me -> peer) lands in peer's bufferDev1 at the offset reserved there for source `me`
        // (recv_offset, fft_mpi_3d_api.cpp:618-625)
        char*       dst = (char*)dstbase + (size_t)x.doffset[peer] * eb;
        const char* src = (const char*)x.sendbuf + (size_t)x.soffset[peer] * eb;
        const size_t bytes = (size_t)x.scount[peer] * eb;
        if (dstdev == mydev) DFFT_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream));
        else DFFT_HIP_TRY(hipMemcpyPeerAsync(dst, dstdev, src, mydev, bytes, stream));
    }
    DFFT_HIP_TRY(hipStreamSynchronize(stream));
    comm_thread_barrier(c);  // all incoming chunks have landed
    return DFFT_OK;
}

static int exchange_rccl(dfft_comm_t c, const ExchangeDesc& x, hipStream_t stream) {
    const size_t eb = elem_bytes(x.dtype);
    const ncclDataType_t ty = x.dtype == DFFT_F64 ? ncclDouble : ncclFloat;
    // self chunk: plain device copy (no link involved)
    if (x.scount[x.me] > 0)
        DFFT_HIP_TRY(hipMemcpyAsync((char*)x.recvbuf + (size_t)x.roffset[x.me] * eb,
                                    (const char*)x.sendbuf + (size_t)x.soffset[x.me] * eb, (size_t)x.scount[x.me] * eb,
                                    hipMemcpyDeviceToDevice, stream));
    if (x.P == 1) return DFFT_OK;
    ncclResult_t r = ncclGroupStart();
    if (r != ncclSuccess) return fail(DFFT_ERCCL, std::string("ncclGroupStart: ") + ncclGetErrorString(r));
    for (int i = 1; i < x.P; ++i) {
        const int to = (x.me + i) % x.P, from = (x.me - i + x.P) % x.P;
        if (x.scount[to] > 0) {
            r = ncclSend((const char*)x.sendbuf + (size_t)x.soffset[to] * eb, (size_t)x.scount[to] * 2, ty, to, c->nccl,
                         stream);
            if (r != ncclSuccess) return fail(DFFT_ERCCL, std::string("ncclSend: ") + ncclGetErrorString(r));
        }
        if (x.rcount[from] > 0) {
            r = ncclRecv((char*)x.recvbuf + (size_t)x.roffset[from] * eb, (size_t)x.rcount[from] * 2, ty, from, c->nccl,
                         stream);
            if (r != ncclSuccess) return fail(DFFT_ERCCL, std::string("ncclRecv: ") + ncclGetErrorString(r));
        }
    }
    r = ncclGroupEnd();
    if (r != ncclSuccess) return fail(DFFT_ERCCL, std::string("ncclGroupEnd: ") + ncclGetErrorString(r));
    return DFFT_OK;
}

int comm_exchange(dfft_comm_t c, const ExchangeDesc& x, hipStream_t stream) {
    if (c->kind == 0) return exchange_local(c, x, stream);
    return exchange_rccl(c, x, stream);
}

}  // namespace dfft

using namespace dfft;

extern "C" {

int dfft_comm_create_local(int total_devices, dfft_comm_t* comm) {
    if (total_devices < 1 || !comm) return fail(DFFT_EINVAL, "dfft_comm_create_local: bad arguments");
    dfft_comm_s* c = new dfft_comm_s;
    c->kind = 0;
    c->P = total_devices;
    c->recvbufs.assign(total_devices, nullptr);
    c->devices.assign(total_devices, 0);
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
    ncclResult_t r = ncclCommInitRank(&c->nccl, total_devices, u, global_idx);
    if (r != ncclSuccess) {
        delete c;
        return fail(DFFT_ERCCL, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
    }
    *comm = c;
    return DFFT_OK;
}

int dfft_comm_destroy(dfft_comm_t comm) {
    if (!comm) return DFFT_OK;
    if (comm->kind == 1 && comm->nccl) ncclCommDestroy(comm->nccl);
    delete comm;
    return DFFT_OK;
}

}  // extern "C"
