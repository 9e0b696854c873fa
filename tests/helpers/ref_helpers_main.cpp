// Test program (tests/test_host_logic.py::test_reference_named_helpers): the helper functions the reference's header makes public
// (3dmpifft_opt/include/fft_mpi_3d_api.h:77-79) called under their reference names through include/fft_mpi_3d_api.h.
//   ref_helpers_main N0 N1 N2 GPU_COUNT MPI_SIZE MPI_RANK   -> "result <total> <in node> <count of local device 0> ... | max <getMaxDataCount>"
#include <cstdio>
#include <cstdlib>

#include <mpi.h>

#include "fft_mpi_3d_api.h"

int main(int argc, char** argv) {
    if (argc != 7) return 2;
    const longInt64 N[3] = {atoll(argv[1]), atoll(argv[2]), atoll(argv[3])};
    const int       ini = atoi(argv[4]), size = atoi(argv[5]), rank = atoi(argv[6]);
    int             total = 0, in_node = 0;
    getProperDeviceNum(N, ini, size, rank, total, in_node);
    longInt64 counts[64] = {0};
    getDataCountForNode(counts, N, rank, size, total, in_node);
    printf("result %d %d", total, in_node);
    for (int i = 0; i < in_node; ++i) printf(" %lld", counts[i]);
    printf(" | max %lld %lld\n", getMaxDataCount((int)N[0], (int)N[1], (int)N[2], total, false), getMaxDataCount((int)N[0], (int)N[1], (int)N[2], total, true));
    return 0;
}
