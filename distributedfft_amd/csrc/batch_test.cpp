// batch_test.cpp -- Test_1D / Test_2D: the reference's single-GPU batched-FFT benchmark programs on top of the C-ABI
// (compile with -DBATCH_DIM=1 or -DBATCH_DIM=2).
//
// Replaces (CLI, protocol, printed lines and CSV schema; not the code) /root/reference/templateFFT/batchTest/Test_1D.cpp
// and Test_2D.cpp, the programs behind the published component tables templateFFT/csv/batch_result{1D,2D}.csv:
//     ./Test_1D X Y Z num_iter printResult      Y is recomputed as 2^26 / X          (Test_1D.cpp:201-209)
//     ./Test_2D X Y Z num_iter printResult      Z is recomputed as 2^26 / (X * Y)    (Test_2D.cpp:196-204)
// i.e. a ~1 GiB fp64 buffer of independent transforms: X is the contiguous axis, a 2D transform is X then Y (stride X).
// Protocol (Test_1D.cpp:29-176): input re = i + 1, im = 0; one warm-up forward transform whose result is kept; num_iter
// timed in-place forward transforms on a second buffer between two HIP events; then the inverse of the kept result and
// max_i |in_i - out_i / (X[*Y])| as "Max error".  Printed: the "FFT: XxYxZ Buffer: .. avg_hip_time: .. Gflops: .." line
// (flops = 5 N log2(X[*Y]), Test_1D.cpp:127-130 / Test_2D.cpp:127-130) and "Max error: ..".
// CSV (the reference keeps its writer under `#if 0`, Test_1D.cpp:178-189; here it is opt-in): with DFFT_BATCH_CSV=<file> one row
//     X,Y,Z,Buffer,hip_time,GFlops,num_iter,bandwidth,max error
// is appended, "bandwidth" exactly as the reference codes it -- buffer / 1.024 * transfers / time with transfers = 2 per
// axis pass for 1D and 4 per axis pass for 2D (Test_2D.cpp:177-184), a formula, not a measured counter.
// The reference's `hipSetDevice(3)` quirk (Test_1D.cpp:55) is not reproduced: device 0, or DFFT_LOCAL_DEVICE.
#include <hip/hip_runtime.h>

#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dfft.h"

#ifndef BATCH_DIM
#error "compile with -DBATCH_DIM=1 or -DBATCH_DIM=2"
#endif

#define HIP_OK(stmt)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (stmt);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            fprintf(stderr, "[%s:%d] %s failed: %s\n", __FILE__, __LINE__, #stmt, hipGetErrorString(e_)); \
            return 2;                                                                                 \
        }                                                                                             \
    } while (0)
#define DFFT_OK_OR_DIE(stmt)                                                               \
    do {                                                                                   \
        int rc_ = (stmt);                                                                  \
        if (rc_ != DFFT_OK) {                                                              \
            fprintf(stderr, "Error! id= %d (%s)\n", rc_, dfft_last_error());               \
            return 3;                                                                      \
        }                                                                                  \
    } while (0)

static int transform(double* buf, long long X, long long Y, long long Z, int dir, hipStream_t s) {
#if BATCH_DIM == 1
    return dfft_fft1d_rows(buf, buf, X, Y * Z, DFFT_F64, dir, s);
#else
    // the plan's t0 stage as a call of its own (dfft_fft2d_batch): planes [Y][X], X contiguous -- Infinity-Cache chunking and, for
    // the plane shapes it is built for, the one-launch stage.  DFFT_BATCH_2D_SEPARATE=1: two whole-buffer 1-D passes (round 5's form)
    static const bool separate = [] {
        const char* e = getenv("DFFT_BATCH_2D_SEPARATE");
        return e && *e == '1';
    }();
    if (!separate) return dfft_fft2d_batch(buf, buf, Y, X, Z, DFFT_F64, dir, s);
    int rc;
    if (dir == DFFT_FORWARD) {
        rc = dfft_fft1d_rows(buf, buf, X, Y * Z, DFFT_F64, dir, s);
        if (rc == DFFT_OK) rc = dfft_fft1d_cols(buf, buf, Y, X, Z, DFFT_F64, dir, s);
    } else {
        rc = dfft_fft1d_cols(buf, buf, Y, X, Z, DFFT_F64, dir, s);
        if (rc == DFFT_OK) rc = dfft_fft1d_rows(buf, buf, X, Y * Z, DFFT_F64, dir, s);
    }
    return rc;
#endif
}

int main(int argc, char* argv[]) {
    if (argc < 6) {
        fprintf(stderr, "usage: %s X Y Z num_iter printResult\n", argv[0]);
        return 1;
    }
    long long X = atoll(argv[1]), Y = atoll(argv[2]), Z = atoll(argv[3]);
    const int num_iter = atoi(argv[4]), printResult = atoi(argv[5]);
    const long long total = 64ll * 32ll * 32768ll;  // 2^26 complex elements = 1 GiB of fp64
#if BATCH_DIM == 1
    if (X < 1) return 1;
    Y = total / X;
    printf("1 - FFT + iFFT C2C 1D in double precision LUT\n");
    const long long norm = X;
#else
    if (X < 1 || Y < 1) return 1;
    Z = total / (X * Y);
    printf("1 - FFT + iFFT C2C 2D in double precision LUT\n");
    const long long norm = X * Y;
#endif
    if (Y < 1 || Z < 1 || num_iter < 1) {
        fprintf(stderr, "nothing to do for %lldx%lldx%lld\n", X, Y, Z);
        return 1;
    }
    const long long N = X * Y * Z;
    if (dfft_device_count() < 1) {
        fprintf(stderr, "no HIP device visible (no CPU fallback)\n");
        return 2;
    }
    const char* dv = getenv("DFFT_LOCAL_DEVICE");
    HIP_OK(hipSetDevice(dv ? atoi(dv) : 0));
    if (!dfft_length_supported(X) || (BATCH_DIM == 2 && !dfft_length_supported(Y))) {
        fprintf(stderr, "Error! unsupported length (7-smooth lengths up to 4096, or products of two tuned lengths above)\n");
        return 3;
    }

    std::vector<double> in(2 * (size_t)N), out(2 * (size_t)N);
    for (long long i = 0; i < N; ++i) {
        in[2 * i] = (double)((int)i) + 1.0;  // Test_1D.cpp:49-52
        in[2 * i + 1] = 0.0;
    }
    const size_t bytes = sizeof(double) * 2 * (size_t)N;
    double *buffer = nullptr, *tmpbuffer = nullptr;
    HIP_OK(hipMalloc((void**)&buffer, bytes));
    HIP_OK(hipMalloc((void**)&tmpbuffer, bytes));
    HIP_OK(hipMemcpy(buffer, in.data(), bytes, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(tmpbuffer, in.data(), bytes, hipMemcpyHostToDevice));
    hipStream_t s = nullptr;  // the reference launches on the NULL stream

    // warm-up forward transform; its result is what the inverse is checked on
    DFFT_OK_OR_DIE(transform(buffer, X, Y, Z, DFFT_FORWARD, s));
    HIP_OK(hipMemcpy(out.data(), buffer, bytes, hipMemcpyDeviceToHost));
    auto print_ends = [&]() {
        for (long long i = 0; i < 8 && i < N; ++i)
            printf("element %lld input:  (%g,%g) output: (%g,%g)\n", i, in[2 * i], in[2 * i + 1], out[2 * i], out[2 * i + 1]);
        for (long long i = (N > 8 ? N - 8 : 0); i < N; ++i)
            printf("element %lld input:  (%g,%g) output: (%g,%g)\n", i, in[2 * i], in[2 * i + 1], out[2 * i], out[2 * i + 1]);
    };
    if (printResult == 1) print_ends();

    hipEvent_t start, stop;
    HIP_OK(hipEventCreate(&start));
    HIP_OK(hipEventCreate(&stop));
    HIP_OK(hipEventRecord(start, s));
    for (int i = 0; i < num_iter; ++i) DFFT_OK_OR_DIE(transform(tmpbuffer, X, Y, Z, DFFT_FORWARD, s));
    HIP_OK(hipEventRecord(stop, s));
    HIP_OK(hipEventSynchronize(stop));
    float elapsed = 0.f;
    HIP_OK(hipEventElapsedTime(&elapsed, start, stop));
    const double avg_ms = elapsed / num_iter;
#if BATCH_DIM == 1
    const double opscount = (double)Y * Z * 5.0 * X * log((double)X) / log(2.0);
    const int    transfers = 2;  // one pass over the buffer: read + write
#else
    const double opscount = 5.0 * (double)N * log((double)X * (double)Y) / log(2.0);
    const int    transfers = 2 * 4;  // two axis passes, 4 "transfers" each as Test_2D.cpp:180 counts them
#endif
    const double mb = bytes / 1024.0 / 1024.0;
    const double gflops = opscount / (1e6 * avg_ms);
    printf("FFT: %lldx%lldx%lld Buffer: %f MB avg_hip_time: %0.6f ms Gflops: %0.6f num_iter: %d \n", X, Y, Z, mb, avg_ms, gflops, num_iter);

    // inverse of the kept forward result
    HIP_OK(hipMemcpy(buffer, out.data(), bytes, hipMemcpyHostToDevice));
    DFFT_OK_OR_DIE(transform(buffer, X, Y, Z, DFFT_BACKWARD, s));
    HIP_OK(hipMemcpy(out.data(), buffer, bytes, hipMemcpyDeviceToHost));
    if (printResult == 1) print_ends();
    double maxErr = 0.0;
    for (long long i = 0; i < N; ++i) {
        const double t1 = in[2 * i] - out[2 * i] / (double)norm, t2 = in[2 * i + 1] - out[2 * i + 1] / (double)norm;
        const double t3 = sqrt(t1 * t1 + t2 * t2);
        maxErr = maxErr >= t3 ? maxErr : t3;
    }
    printf("Max error: %g\n", maxErr);

    if (const char* csv = getenv("DFFT_BATCH_CSV")) {
        FILE* f = fopen(csv, "a");
        if (f) {
            fprintf(f, "%lld,%lld,%lld,%g,%g,%g,%d,%g,%g\n", X, Y, Z, mb, avg_ms, gflops, num_iter, mb / 1.024 * transfers / avg_ms, maxErr);
            fclose(f);
        }
    }
    (void)hipFree(buffer);
    (void)hipFree(tmpbuffer);
    (void)hipEventDestroy(start);
    (void)hipEventDestroy(stop);
    return 0;
}
