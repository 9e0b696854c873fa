// ref_dump_driver.cpp -- TEST INFRASTRUCTURE.  Our own small driver that calls the REFERENCE's distributed-FFT API
// (compiled from /root/reference/3dmpifft_opt/include/*.cpp + templateFFT/src/templateFFT.cpp by oracle/Makefile) to
// obtain the reference's forward output and per-stage times for a given size on this GPU.  It is the "reference itself
// run here" oracle for P = 1 (gpurun exposes one GPU).  Nothing from the reference is copied: this file only uses the
// public functions declared in fft_mpi_3d_api.h:68-79.
//
//   distFFTOpt_ref NX NY NZ [input] [dumpfile] [reps]
//     input   : "index" (re = im = linear index, the reference driver's input, fftSpeed3d_c2c.cpp:62)
//               or a path to raw fp64 (re,im) pairs of NX*NY*NZ elements
//     dumpfile: forward result [NY][NZ][NX] (kx fastest) as raw fp64 pairs
//     reps    : number of timed forward executes (each prints the reference's own "t0: ..." line)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fft_mpi_3d_api.h"

int main(int argc, char** argv) {
    int provided;
    MPI_CHECK(MPI_Init_thread(&argc, &argv, MPI_THREAD_SERIALIZED, &provided));
    if (argc < 4) {
        fprintf(stderr, "usage: %s NX NY NZ [index|file] [dumpfile] [reps]\n", argv[0]);
        return 2;
    }
    const longInt64   N[3] = {atoll(argv[1]), atoll(argv[2]), atoll(argv[3])};
    const std::string input = argc > 4 ? argv[4] : "index";
    const char*       dump = argc > 5 ? argv[5] : nullptr;
    const int         reps = argc > 6 ? atoi(argv[6]) : 3;

    int       newDeviceCount, newDeviceCountInNode;
    longInt64 dataCountInNode[1];
    fft_mpi_init(N, 1, MPI_COMM_WORLD, newDeviceCount, newDeviceCountInNode, dataCountInNode);
    ROCM_CHECK(hipSetDevice(0));
    const size_t        count = (size_t)dataCountInNode[0];
    std::vector<double> host(2 * count);
    if (input == "index") {
        for (size_t j = 0; j < count; ++j) host[2 * j] = host[2 * j + 1] = (double)j;
    } else {
        FILE* f = fopen(input.c_str(), "rb");
        if (!f || fread(host.data(), sizeof(double), host.size(), f) != host.size()) {
            fprintf(stderr, "cannot read %s\n", input.c_str());
            return 3;
        }
        fclose(f);
    }
    const longInt64 maxCount = getMaxDataCount(N[0], N[1], N[2], 1, true);
    Complex *       inDev = fft_mpi_alloc_local_memory(maxCount, ALLOC_DEV), *outDev = fft_mpi_alloc_local_memory(maxCount, ALLOC_DEV);
    Complex*        node_data[1];
    ROCM_CHECK(hipMemcpy(inDev, host.data(), count * sizeof(Complex), hipMemcpyHostToDevice));
    fft_mpi_3d_plan_p plan = fft_mpi_plan_dft_c2c_3d(N[0], N[1], N[2], inDev, outDev, node_data, MPI_COMM_WORLD, 0, 1, 1, FORWARD);
    ROCM_CHECK(hipMemcpy(plan->bufferDev1, host.data(), count * sizeof(Complex), hipMemcpyHostToDevice));
    fft_mpi_execute_dft_3d_c2c(plan);
    if (dump) {
        std::vector<double> out(2 * count);
        ROCM_CHECK(hipMemcpy(out.data(), outDev, count * sizeof(Complex), hipMemcpyDeviceToHost));
        FILE* f = fopen(dump, "wb");
        if (!f || fwrite(out.data(), sizeof(double), out.size(), f) != out.size()) {
            fprintf(stderr, "cannot write %s\n", dump);
            return 4;
        }
        fclose(f);
    }
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        ROCM_CHECK(hipMemcpy(plan->bufferDev1, host.data(), count * sizeof(Complex), hipMemcpyHostToDevice));
        double t = -MPI_Wtime();
        fft_mpi_execute_dft_3d_c2c(plan);
        t += MPI_Wtime();
        if (t < best) best = t;
    }
    const double n = (double)N[0] * N[1] * N[2];
    printf("REF_FORWARD_BEST_S %.9f\nREF_GFLOPS %.3f\n", best, 5.0 * n * log2(n) * 1e-9 / best);
    fft_mpi_destroy_plan(plan);
    ROCM_CHECK(hipFree(inDev));
    ROCM_CHECK(hipFree(outDev));
    MPI_CHECK(MPI_Finalize());
    return 0;
}
