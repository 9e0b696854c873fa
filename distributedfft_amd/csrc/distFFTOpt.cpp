// distFFTOpt.cpp -- benchmark + self-check driver of the MI355X-native slab FFT.
//
// Keeps the command line, the printed lines and the error/throughput formulas of the reference's driver
// (/root/reference/3dmpifft_opt/fftSpeed3d_c2c.cpp: argv :28-37, input :56-63, round-trip error :84-91, timed forward
// :94-98, report :126-138) so scripts written against it keep working, but is organised differently: one worker per local
// GPU (std::thread, not an OpenMP team), 64-bit element counts, results gathered in a struct and reduced at the end.
//
//   distFFTOpt NX NY NZ GPU_COUNT        GPU_COUNT = GPUs driven by this process (one worker thread each)
//   distFFTOpt --device-count            helper for speedTest.sh
//
// Launch: `sh speedTest.sh <ranks> X Y Z` starts <ranks> processes with one GPU each (RCCL between them); a real
// `mpirun -np k` also works when the build uses a real <mpi.h>.  Environment knobs (none changes the printed surface):
//   DFFT_TIMED_REPS=k   time k forward executes and report the best (default 1 = the reference's single sample)
//   DFFT_DUMP=path      forward result of device g written to path.<g> (raw fp64 pairs) for parity tests
//   DFFT_LOCAL_DEVICE=d bind this process to HIP device d (set per rank by speedTest.sh)
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fft_mpi_3d_api.h"

namespace {

struct Options {
    longInt64   n[3] = {0, 0, 0};
    int         gpus_per_process = 0;
    int         timed_reps = 1;
    const char* dump_prefix = nullptr;
};

struct Layout {  // what fft_mpi_init decided
    int                    total_devices = 0, local_devices = 0;
    std::vector<longInt64> local_counts;
};

struct Outcome {  // per process, max over its devices
    double max_error = 1e-30;
    double forward_seconds = 1e-30;
};

[[noreturn]] void usage_and_exit() {
    printf("The format of arguments should be [NX, NY, NZ, GPU_COUNT]!\n");  // fftSpeed3d_c2c.cpp:29
    exit(EXIT_FAILURE);
}

Options parse(int argc, char** argv) {
    if (argc != 5) usage_and_exit();
    Options o;
    for (int i = 0; i < 3; ++i) o.n[i] = atoll(argv[1 + i]);
    o.gpus_per_process = atoi(argv[4]);
    if (o.gpus_per_process < 1) usage_and_exit();
    if (const char* r = getenv("DFFT_TIMED_REPS"))
        if (atoi(r) > 0) o.timed_reps = atoi(r);
    o.dump_prefix = getenv("DFFT_DUMP");
    return o;
}

// value(j) = global linear index in both components (fftSpeed3d_c2c.cpp:59-63)
void fill_index_ramp(std::vector<double>& host, longInt64 first_index) {
    const size_t count = host.size() / 2;
    for (size_t j = 0; j < count; ++j) host[2 * j] = host[2 * j + 1] = (double)(first_index + (longInt64)j);
}

// max_j |x_j - y_j / N| / 1e7, the metric the reference prints as "Max error" (fftSpeed3d_c2c.cpp:84-91)
double roundtrip_error(const std::vector<double>& x, const std::vector<double>& y, double total_elements) {
    double worst = -1.0;
    for (size_t j = 0; j + 1 < x.size(); j += 2) {
        const double dr = x[j] - y[j] / total_elements, di = x[j + 1] - y[j + 1] / total_elements;
        worst = std::max(worst, std::sqrt(dr * dr + di * di) / 1e7);
    }
    return worst;
}

void dump_forward(const Options& o, const Layout& lay, int global_dev, const Complex* out_dev) {
    longInt64 rows = 0;
    DFFT_CHECK(dfft_local_size(o.n[0], o.n[1], o.n[2], lay.total_devices, global_dev, nullptr, nullptr, &rows, nullptr));
    std::vector<double> host(2 * (size_t)(rows * o.n[2] * o.n[0]));
    ROCM_CHECK(hipMemcpy(host.data(), out_dev, host.size() * sizeof(double), hipMemcpyDeviceToHost));
    const std::string path = std::string(o.dump_prefix) + "." + std::to_string(global_dev);
    FILE*             f = fopen(path.c_str(), "wb");
    if (!f || fwrite(host.data(), sizeof(double), host.size(), f) != host.size()) {
        fprintf(stderr, "cannot write %s\n", path.c_str());
        exit(EXIT_FAILURE);
    }
    fclose(f);
}

// Everything one device does: plan both directions, round-trip check, timed forward.
void run_device(const Options& o, const Layout& lay, int local_dev, int mpi_rank, int mpi_size, int visible_devices,
                std::vector<Complex*>& node_data, Outcome& shared, std::mutex& guard) {
    const int global_dev = mpi_rank * (int)std::ceil((double)lay.total_devices / mpi_size) + local_dev;
    const char* bound = getenv("DFFT_LOCAL_DEVICE");
    ROCM_CHECK(hipSetDevice((bound ? atoi(bound) : global_dev) % visible_devices));  // fftSpeed3d_c2c.cpp:53

    const longInt64 planes_per_device = (longInt64)std::ceil((double)o.n[0] / lay.total_devices);
    const longInt64 count = lay.local_counts[local_dev];
    const size_t    bytes = (size_t)count * sizeof(Complex);
    std::vector<double> input(2 * (size_t)count), back(2 * (size_t)count);
    fill_index_ramp(input, (longInt64)global_dev * planes_per_device * o.n[1] * o.n[2]);

    const bool      last = global_dev == lay.total_devices - 1;
    const longInt64 capacity = getMaxDataCount((int)o.n[0], (int)o.n[1], (int)o.n[2], lay.total_devices, last);
    Complex*        in_dev = (Complex*)dfft_alloc(capacity, DFFT_F64, ALLOC_DEV);
    Complex*        out_dev = (Complex*)dfft_alloc(capacity, DFFT_F64, ALLOC_DEV);
    if (!in_dev || !out_dev) {
        fprintf(stderr, "device allocation failed: %s\n", dfft_last_error());
        exit(EXIT_FAILURE);
    }
    ROCM_CHECK(hipMemset(in_dev, 0, (size_t)capacity * sizeof(Complex)));
    ROCM_CHECK(hipMemcpy(in_dev, input.data(), bytes, hipMemcpyHostToDevice));

    fft_mpi_3d_plan_p fwd = fft_mpi_plan_dft_c2c_3d(o.n[0], o.n[1], o.n[2], in_dev, out_dev, node_data.data(), MPI_COMM_WORLD,
                                                    local_dev, lay.local_devices, lay.total_devices, FORWARD);
    auto reload = [&] { ROCM_CHECK(hipMemcpy(fwd->bufferDev1, input.data(), bytes, hipMemcpyHostToDevice)); };  // :78
    reload();
    fft_mpi_execute_dft_3d_c2c(fwd);
    if (o.dump_prefix) dump_forward(o, lay, global_dev, out_dev);

    fft_mpi_3d_plan_p bwd = fft_mpi_plan_dft_c2c_3d(o.n[0], o.n[1], o.n[2], out_dev, in_dev, node_data.data(), MPI_COMM_WORLD,
                                                    local_dev, lay.local_devices, lay.total_devices, BACKWARD);
    fft_mpi_execute_dft_3d_c2c(bwd);
    ROCM_CHECK(hipMemcpy(back.data(), in_dev, bytes, hipMemcpyDeviceToHost));
    const double err = roundtrip_error(input, back, (double)o.n[0] * (double)o.n[1] * (double)o.n[2]);

    // Warm-up, then the timed forward (:94-98).  A forward execute consumes bufferDev1, so the input is reloaded before
    // every execute: each timed run transforms the real input, not the leftovers of the previous one.
    reload();
    fft_mpi_execute_dft_3d_c2c(fwd);
    double best = 1e30;
    for (int r = 0; r < o.timed_reps; ++r) {
        reload();
        if (lay.local_devices == 1) MPI_CHECK(MPI_Barrier(MPI_COMM_WORLD));  // the control plane is single-threaded
        const double t0 = MPI_Wtime();
        fft_mpi_execute_dft_3d_c2c(fwd);
        best = std::min(best, MPI_Wtime() - t0);
    }

    fft_mpi_destroy_plan(fwd);
    fft_mpi_destroy_plan(bwd);
    DFFT_CHECK(dfft_free(in_dev, ALLOC_DEV));
    DFFT_CHECK(dfft_free(out_dev, ALLOC_DEV));
    std::lock_guard<std::mutex> lk(guard);
    shared.max_error = std::max(shared.max_error, err);
    shared.forward_seconds = std::max(shared.forward_seconds, best);
}

void print_report(const Options& o, int mpi_size, double seconds, double max_error) {  // fftSpeed3d_c2c.cpp:126-138
    const long long elements = o.n[0] * o.n[1] * o.n[2];
    const double    gflops = 5.0 * elements * std::log((double)elements) * 1e-9 / std::log(2.0) / seconds;
    std::cout << "\n----------------------------------------------------------------------------- \n";
    std::cout << "distributed FFT performance test\n";
    std::cout << "----------------------------------------------------------------------------- \n";
    std::cout << "Size:             " << o.n[0] << "x" << o.n[1] << "x" << o.n[2] << "\n";
    std::cout << "MPI ranks:        " << mpi_size << "\n";
    std::cout << "Forward FFT time: " << seconds << " (s)\n";
    std::cout << "Performance:      " << gflops << " GFlops/s\n";
    std::cout << "Max error:        " << max_error << "\n";
    std::cout << std::endl;
}

}  // namespace

int main(int argc, char* argv[]) {
    if (argc == 2 && std::string(argv[1]) == "--device-count") {
        printf("%d\n", dfft_device_count());
        return 0;
    }
    char host[256];
    gethostname(host, sizeof(host));
    printf("PID %d on %s ready for attach\n", getpid(), host);  // fftSpeed3d_c2c.cpp:13
    fflush(stdout);

    int provided = 0;
    MPI_CHECK(MPI_Init_thread(&argc, &argv, MPI_THREAD_SERIALIZED, &provided));
    if (provided != MPI_THREAD_SERIALIZED) {
        printf("could not support multi-thread MPI!\n");
        exit(EXIT_FAILURE);
    }
    int mpi_size = 1, mpi_rank = 0;
    MPI_CHECK(MPI_Comm_size(MPI_COMM_WORLD, &mpi_size));
    MPI_CHECK(MPI_Comm_rank(MPI_COMM_WORLD, &mpi_rank));
    const Options opt = parse(argc, argv);

    int visible = 0;
    ROCM_CHECK(hipGetDeviceCount(&visible));
    if (visible < 1) {
        fprintf(stderr, "no HIP device visible: this library has no CPU fallback\n");
        exit(EXIT_FAILURE);
    }

    Layout lay;
    lay.local_counts.resize(opt.gpus_per_process);
    fft_mpi_init(opt.n, opt.gpus_per_process, MPI_COMM_WORLD, lay.total_devices, lay.local_devices, lay.local_counts.data());

    std::vector<Complex*>    node_data(lay.local_devices, nullptr);
    Outcome                  mine;
    std::mutex               guard;
    std::vector<std::thread> workers;
    for (int d = 1; d < lay.local_devices; ++d)
        workers.emplace_back(run_device, std::cref(opt), std::cref(lay), d, mpi_rank, mpi_size, visible, std::ref(node_data),
                             std::ref(mine), std::ref(guard));
    run_device(opt, lay, 0, mpi_rank, mpi_size, visible, node_data, mine, guard);
    for (auto& w : workers) w.join();
    MPI_CHECK(MPI_Barrier(MPI_COMM_WORLD));

    Outcome all;
    MPI_CHECK(MPI_Reduce(&mine.max_error, &all.max_error, 1, MPI_DOUBLE, MPI_MAX, 0, MPI_COMM_WORLD));
    MPI_CHECK(MPI_Reduce(&mine.forward_seconds, &all.forward_seconds, 1, MPI_DOUBLE, MPI_MAX, 0, MPI_COMM_WORLD));
    if (mpi_rank == 0) print_report(opt, mpi_size, all.forward_seconds, all.max_error);

    fft_mpi_cleanup();
    MPI_CHECK(MPI_Finalize());
    return 0;
}
