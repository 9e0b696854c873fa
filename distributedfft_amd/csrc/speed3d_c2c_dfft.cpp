// speed3d_c2c_dfft.cpp -- the heFFTe `speed3d_c2c` benchmark protocol on top of libdfft_mi355x, so that the GPU path is
// timed with exactly the methodology that times the CPU baseline (the heFFTe 2.1.0 harness bundled with the reference:
// /root/reference/heffte/heffteBenchmark/benchmarks/speed3d.h:84-183, CLI :203-257).  SURVEY section 8(f)-4.
//
//   speed3d_c2c <backend> <double|float> <X> <Y> <Z> [heFFTe options, ignored]
//
// Protocol reproduced: input = std::minstd_rand(4242) uniform(0,1) per local element (test_fft3d.h:20-28); identical in
// and out distribution (here: natural X slabs, DFFT_PLAN_NATURAL); warm-up forward+backward; ntest = 5 iterations of
// { forward with full 1/N scaling ; backward }; "Time per run" = t / (2*ntest) (speed3d.h:109-117,157); max |in - out|
// against the tolerance 1e-11 (double) / 5e-4 (float) (test_common.h:136-140); GFlops = 5 N log2 N / t (speed3d.h:159).
// heFFTe's index 0 is the fastest dimension, so X = our N2, Y = N1, Z = N0.
// Multi-process: one process per GPU, rendezvous through dfft_boot_* (DFFT_RANK / DFFT_WORLD_SIZE / ... or a launcher's
// PMI/OMPI variables), data plane RCCL.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <random>
#include <string>
#include <vector>

#include "dfft.h"

#define CHECK_DFFT(stmt)                                                                                  \
    do {                                                                                                  \
        int rc_ = (stmt);                                                                                 \
        if (rc_ != DFFT_OK) {                                                                             \
            fprintf(stderr, "[%s:%d] '%s' failed with %d: %s\n", __FILE__, __LINE__, #stmt, rc_, dfft_last_error()); \
            exit(EXIT_FAILURE);                                                                           \
        }                                                                                                 \
    } while (0)
#define CHECK_HIP(stmt)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (stmt);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            fprintf(stderr, "[%s:%d] '%s' failed: %s\n", __FILE__, __LINE__, #stmt, hipGetErrorString(e_)); \
            exit(EXIT_FAILURE);                                                                           \
        }                                                                                                 \
    } while (0)

template <class Real> static int run(long long n0, long long n1, long long n2, int dtype, double tolerance) {
    const int me = dfft_boot_rank(), nprocs = dfft_boot_size();
    int       visible = 0;
    CHECK_HIP(hipGetDeviceCount(&visible));
    if (visible < 1) {
        fprintf(stderr, "no HIP device visible: this library has no CPU fallback\n");
        return 1;
    }
    const char* bound = getenv("DFFT_LOCAL_DEVICE");
    CHECK_HIP(hipSetDevice((bound ? atoi(bound) : me) % visible));

    const long long N[3] = {n0, n1, n2};
    int             total = 0, in_rank = 0;
    CHECK_DFFT(dfft_proper_device_count(N, 1, nprocs, me, -1, &total, &in_rank));
    if (total != nprocs) {
        if (me == 0) fprintf(stderr, "%lld planes cannot be cut into %d slabs\n", n0, nprocs);
        return 1;
    }
    dfft_comm_t comm = nullptr;
    if (nprocs > 1) {
        const char* ex = getenv("DFFT_EXCHANGE");
        if (ex && (std::string(ex) == "ipc" || std::string(ex) == "ipc-async")) {  // hipIpc peer copies: no RCCL
            CHECK_DFFT(dfft_comm_create_ipc(nprocs, me, std::string(ex) == "ipc-async" ? 1 : 0, &comm));
        } else {
            char id[128];
            if (me == 0) CHECK_DFFT(dfft_rccl_unique_id(id));
            CHECK_DFFT(dfft_boot_bcast(id, sizeof(id), 0));
            CHECK_DFFT(dfft_comm_create_rccl(id, nprocs, me, &comm));
        }
    }
    const long long count = dfft_local_count(N, nprocs, me);
    const long long cap = dfft_max_count(n0, n1, n2, nprocs, me == nprocs - 1);
    const size_t    esz = 2 * sizeof(Real);

    // input: the same generator and seed on every rank, one draw per local element, real part only
    std::minstd_rand                       park_miller(4242);
    std::uniform_real_distribution<double> unif(0.0, 1.0);
    std::vector<Real>                      input(2 * (size_t)count, Real(0)), output(2 * (size_t)count);
    for (long long i = 0; i < count; ++i) input[2 * i] = static_cast<Real>(unif(park_miller));

    void *a = dfft_alloc(cap, dtype, DFFT_ALLOC_DEV), *b = dfft_alloc(cap, dtype, DFFT_ALLOC_DEV),
         *c = dfft_alloc(cap, dtype, DFFT_ALLOC_DEV);
    if (!a || !b || !c) {
        fprintf(stderr, "allocation failed: %s\n", dfft_last_error());
        return 1;
    }
    CHECK_HIP(hipMemset(a, 0, (size_t)cap * esz));
    CHECK_HIP(hipMemcpy(a, input.data(), (size_t)count * esz, hipMemcpyHostToDevice));
    const unsigned flags = DFFT_PLAN_NATURAL | DFFT_PLAN_INPUT_FROM_IN;
    dfft_plan_t    fwd = nullptr, bwd = nullptr;
    CHECK_DFFT(dfft_plan_create(&fwd, n0, n1, n2, dtype, DFFT_FORWARD, a, b, comm, me, nprocs, flags));
    CHECK_DFFT(dfft_plan_create(&bwd, n0, n1, n2, dtype, DFFT_BACKWARD, b, c, comm, me, nprocs, flags));
    // scale::full on the forward transform: folded into the X-pass kernel, no extra pass over the data
    CHECK_DFFT(dfft_plan_set_scale(fwd, 1.0 / ((double)n0 * (double)n1 * (double)n2)));
    auto round_trip = [&] {
        CHECK_DFFT(dfft_execute(fwd, DFFT_EXEC_NO_TIMING));
        CHECK_DFFT(dfft_plan_sync(fwd));
        CHECK_DFFT(dfft_execute(bwd, DFFT_EXEC_NO_TIMING));
        CHECK_DFFT(dfft_plan_sync(bwd));
    };
    round_trip();  // warm-up (speed3d.h:105-106)
    const int ntest = 5;
    CHECK_DFFT(dfft_boot_barrier());
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < ntest; ++i) round_trip();
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_DFFT(dfft_boot_barrier());
    double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    CHECK_DFFT(dfft_boot_allreduce_max(&t, 1));

    CHECK_HIP(hipMemcpy(output.data(), c, (size_t)count * esz, hipMemcpyDeviceToHost));
    double err = 0.0;
    for (long long i = 0; i < count; ++i) {
        const double dr = (double)input[2 * i] - (double)output[2 * i], di = (double)input[2 * i + 1] - (double)output[2 * i + 1];
        err = std::max(err, std::sqrt(dr * dr + di * di));
    }
    CHECK_DFFT(dfft_boot_allreduce_max(&err, 1));
    if (err > tolerance) {
        if (me == 0)
            std::cout << "------------------------------- \n"
                      << "ERROR: observed error after heFFTe benchmark exceeds the tolerance\n"
                      << "       tolerance: " << tolerance << "  error: " << err << std::endl;
        return 1;
    }
    if (me == 0) {
        const double t_run = t / (2.0 * ntest);
        const double fftsize = (double)n0 * (double)n1 * (double)n2;
        const double floprate = 5.0 * fftsize * std::log(fftsize) * 1e-9 / std::log(2.0) / t_run;
        const long long mem_mb = 4ll * cap * (long long)esz / (1024ll * 1024ll);  // in, out, bufferDev1 (+ receive buffer)
        std::cout << "\n----------------------------------------------------------------------------- \n";
        std::cout << "heFFTe performance test\n";
        std::cout << "----------------------------------------------------------------------------- \n";
        std::cout << "Backend:   dfft-mi355x (libdfft_mi355x.so, gfx950)\n";
        std::cout << "Size:      " << n2 << "x" << n1 << "x" << n0 << "\n";
        std::cout << "MPI ranks: " << std::setw(4) << nprocs << "\n";
        std::cout << "Grids: (1, 1, " << nprocs << ")  (1, " << nprocs << ", 1)  (1, 1, " << nprocs << ")  \n";
        std::cout << "Time per run: " << t_run << " (s)\n";
        std::cout << "Performance:  " << floprate << " GFlops/s\n";
        std::cout << "Memory usage: " << mem_mb << "MB/rank\n";
        std::cout << "Tolerance:    " << tolerance << "\n";
        std::cout << "Max error:    " << err << "\n";
        std::cout << std::endl;
    }
    dfft_plan_destroy(fwd);
    dfft_plan_destroy(bwd);
    dfft_comm_destroy(comm);
    dfft_free(a, DFFT_ALLOC_DEV);
    dfft_free(b, DFFT_ALLOC_DEV);
    dfft_free(c, DFFT_ALLOC_DEV);
    return 0;
}

int main(int argc, char** argv) {
    CHECK_DFFT(dfft_boot_init());
    if (argc < 6) {
        if (dfft_boot_rank() == 0)
            std::cout << "\nUsage:\n    speed3d_c2c <backend> <precision> <size-x> <size-y> <size-z> <args>\n\n"
                      << "    backend is ignored (always the MI355X-native library), precision is float or double,\n"
                      << "    heFFTe's reshape/decomposition options are accepted and ignored (slabs + all-to-all always).\n";
        dfft_boot_finalize();
        return 0;
    }
    const std::string precision = argv[2];
    const long long   x = atoll(argv[3]), y = atoll(argv[4]), z = atoll(argv[5]);
    int               rc;
    if (precision == "double" || precision == "double-long") rc = run<double>(z, y, x, DFFT_F64, 1e-11);
    else if (precision == "float" || precision == "float-long") rc = run<float>(z, y, x, DFFT_F32, 5e-4);
    else {
        if (dfft_boot_rank() == 0) std::cout << "Invalid precision!\nMust use float or double" << std::endl;
        rc = 0;
    }
    dfft_boot_finalize();
    return rc;
}
