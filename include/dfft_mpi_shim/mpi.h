// mpi.h (shim) -- the handful of MPI calls the DistributedFFT driver makes, mapped onto libdfft_mi355x's TCP
// rendezvous (dfft_boot_*, include/dfft.h).  Put this directory on the include path ONLY when no real MPI is wanted:
// the data plane (t2) never goes through MPI in this library (RCCL over xGMI), so MPI is needed for rank/size,
// one broadcast, barriers and a MAX reduction (fftSpeed3d_c2c.cpp:18-26, 120-124) -- nothing else.
#ifndef DFFT_MPI_SHIM_H
#define DFFT_MPI_SHIM_H

#include <chrono>
#include <cstring>

#include "../dfft.h"

typedef int MPI_Comm;
typedef int MPI_Datatype;
typedef int MPI_Op;

#define MPI_COMM_WORLD 0
#define MPI_SUCCESS 0
#define MPI_ERR_OTHER 15
#define MPI_THREAD_SINGLE 0
#define MPI_THREAD_FUNNELED 1
#define MPI_THREAD_SERIALIZED 2
#define MPI_THREAD_MULTIPLE 3
#define MPI_BYTE 1
#define MPI_INT 4
#define MPI_DOUBLE 8
#define MPI_MAX 100

static inline int MPI_Init_thread(int*, char***, int required, int* provided) {
    if (provided) *provided = required;
    return dfft_boot_init() == DFFT_OK ? MPI_SUCCESS : MPI_ERR_OTHER;
}
static inline int MPI_Init(int* a, char*** b) {
    int p;
    return MPI_Init_thread(a, b, MPI_THREAD_SINGLE, &p);
}
static inline int MPI_Comm_size(MPI_Comm, int* size) {
    *size = dfft_boot_size();
    return MPI_SUCCESS;
}
static inline int MPI_Comm_rank(MPI_Comm, int* rank) {
    *rank = dfft_boot_rank();
    return MPI_SUCCESS;
}
static inline double MPI_Wtime(void) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static inline int MPI_Barrier(MPI_Comm) { return dfft_boot_barrier() == DFFT_OK ? MPI_SUCCESS : MPI_ERR_OTHER; }
static inline int MPI_Bcast(void* buf, int count, MPI_Datatype type, int root, MPI_Comm) {
    return dfft_boot_bcast(buf, (size_t)count * (size_t)type, root) == DFFT_OK ? MPI_SUCCESS : MPI_ERR_OTHER;
}
// Only MPI_DOUBLE + MPI_MAX (the driver's two reductions).
static inline int MPI_Reduce(const void* sendbuf, void* recvbuf, int count, MPI_Datatype type, MPI_Op op, int root,
                             MPI_Comm) {
    if (type != MPI_DOUBLE || op != MPI_MAX || count < 1 || count > 64) return MPI_ERR_OTHER;
    double tmp[64];
    std::memcpy(tmp, sendbuf, sizeof(double) * (size_t)count);
    if (dfft_boot_allreduce_max(tmp, count) != DFFT_OK) return MPI_ERR_OTHER;
    if (dfft_boot_rank() == root && recvbuf) std::memcpy(recvbuf, tmp, sizeof(double) * (size_t)count);
    return MPI_SUCCESS;
}
static inline int MPI_Finalize(void) { return dfft_boot_finalize() == DFFT_OK ? MPI_SUCCESS : MPI_ERR_OTHER; }

#endif  // DFFT_MPI_SHIM_H
