// fft_mpi_common.h -- types, constants and CHECK macros of the reference's public surface, re-stated for the
// MI355X-native library (drop-in for /root/reference/3dmpifft_opt/include/fft_mpi_common.h:15-103).
// Same names and meaning; none of the reference's rocFFT / hipFFT / hiprtc / templateFFT dependencies.
#ifndef __FFT_MPI_COMMON_H__
#define __FFT_MPI_COMMON_H__

#include <hip/hip_runtime.h>
#include <mpi.h>  // a real MPI, or include/dfft_mpi_shim/mpi.h (TCP rendezvous, no MPI installation needed)

#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "dfft.h"

#define ALLOC_CPU DFFT_ALLOC_HOST   // fft_mpi_common.h:15
#define ALLOC_DEV DFFT_ALLOC_DEV    // :16
#define FORWARD DFFT_FORWARD        // :18
#define BACKWARD DFFT_BACKWARD      // :19

typedef double    Complex[2];  // :21  interleaved re, im
typedef long long longInt64;   // :22

struct TransInfo {  // :24-29, element counts/offsets per peer
    longInt64* soffset;
    longInt64* scount;
    longInt64* roffset;
    longInt64* rcount;
};

// Error behaviour of the reference: print "[file:line] ... failed" to stderr and exit(EXIT_FAILURE) (:31-103).
#define MPI_CHECK(stmt)                                                                        \
    do {                                                                                       \
        int mpi_errno = (stmt);                                                                \
        if (MPI_SUCCESS != mpi_errno) {                                                        \
            fprintf(stderr, "[%s:%d] MPI call failed with %d \n", __FILE__, __LINE__, mpi_errno); \
            exit(EXIT_FAILURE);                                                                \
        }                                                                                      \
    } while (0)

#define ROCM_CHECK(stmt)                                                                                    \
    do {                                                                                                    \
        hipError_t rocm_errno = (stmt);                                                                     \
        if (0 != rocm_errno) {                                                                              \
            fprintf(stderr, "[%s:%d] ROCM call '%s' failed with %d: %s \n", __FILE__, __LINE__, #stmt,      \
                    rocm_errno, hipGetErrorString(rocm_errno));                                             \
            exit(EXIT_FAILURE);                                                                             \
        }                                                                                                   \
    } while (0)

// Replaces OPTFFT_CHECK / ROCFFT_CHECK / HIPFFT_CHECK / NCCLCHECK: one macro for the dfft C-ABI return codes.
#define DFFT_CHECK(stmt)                                                                                    \
    do {                                                                                                    \
        int dfft_errno = (stmt);                                                                            \
        if (DFFT_OK != dfft_errno) {                                                                        \
            fprintf(stderr, "[%s:%d] dfft call '%s' failed with %d: %s\n", __FILE__, __LINE__, #stmt,       \
                    dfft_errno, dfft_last_error());                                                         \
            exit(EXIT_FAILURE);                                                                             \
        }                                                                                                   \
    } while (0)

#endif  // __FFT_MPI_COMMON_H__
