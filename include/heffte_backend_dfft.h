/* heffte_backend_dfft.h -- heFFTe 1-D executor backend on the MI355X kernels of libdfft_mi355x.
 *
 * heFFTe (bundled with the reference under heffte/heffteBenchmark, version 2.1.0) builds a distributed 3D FFT out of
 * MPI reshapes and a per-backend "one-dimensional executor": a class that transforms all 1-D lines of a local box along one
 * dimension.  This header provides that executor on top of the C-ABI's batched building blocks (dfft_fft1d_rows /
 * dfft_fft1d_cols, include/dfft.h), so that heFFTe's own benchmark and tests -- benchmarks/speed3d_c2c.cpp, test/test_fft3d.h,
 * tolerance 1e-11 / 5e-4 (test_common.h:136-140) -- run their plan logic, packing and MPI exchange unchanged while every
 * FFT is computed by the gfx950 kernels.
 *
 * It takes the PLACE of heFFTe's stock backend header: it defines that header's include guard and supplies the same
 * specialisations for the `heffte::backend::stock` tag that heffte_backend_stock.h provides
 *     backend::is_enabled<stock>                      heffte_backend_stock.h:28
 *     stock_fft_executor  -> dfft_fft_executor        heffte_backend_stock.h:431-510   (constructor geometry :434-443,
 *                                                     forward/backward on std::complex<float|double> :446-473, box_size :499)
 *     one_dim_backend<backend::stock>                 heffte_backend_stock.h:595-627
 *     default_plan_options<backend::stock>            heffte_backend_stock.h:633-636
 * so no file of heFFTe has to be edited: compile heFFTe's sources and the benchmark with
 *     -include <this header>          (before anything else; it pulls heffte_pack3d.h exactly like the header it replaces)
 * and `speed3d_c2c stock double X Y Z ...` then runs on the GPU kernels (recipe: oracle/Makefile, target
 * _ref/speed3d_c2c_dfft; test: tests/test_gpu_heffte_backend.py).  The stock tag describes host memory, so the executor
 * receives host arrays: it keeps one device buffer per executor, copies the box in, transforms in place, copies it back.
 * That makes it a functional shim (the reshapes stay on the CPU and every 1-D stage crosses PCIe twice), not the fast path
 * -- the fast path is the slab pipeline behind dfft_plan_create.  Real-to-complex executors are outside the hot path
 * (DESIGN.md section 8): they are declared so that heFFTe's r2c translation unit still compiles, and throw when used.
 */
#ifndef HEFFTE_BACKEND_STOCK_FFT_H
#define HEFFTE_BACKEND_STOCK_FFT_H /* this header replaces heffte_backend_stock.h */
#define HEFFTE_BACKEND_DFFT_H

#include <hip/hip_runtime_api.h>

#include <complex>
#include <memory>
#include <stdexcept>
#include <string>

#include "dfft.h"
#include "heffte_pack3d.h"

namespace heffte {

namespace backend {
template <> struct is_enabled<stock> : std::true_type {};
}  // namespace backend

namespace dfft_shim {
inline void hip_ok(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string("heffte dfft backend: ") + what + ": " + hipGetErrorString(e));
}
inline void dfft_ok(int rc, const char* what) {
    if (rc != DFFT_OK) throw std::runtime_error(std::string("heffte dfft backend: ") + what + ": " + dfft_last_error());
}
}  // namespace dfft_shim

/* All 1-D transforms of a box along `dimension`, in place, unnormalised (heFFTe scales separately).  With o0, o1, o2 the box
 * extents from the fastest to the slowest index (box.osize): the lines are contiguous rows (dimension == order[0]), the
 * columns of o2 matrices [o1][o0] (order[1]) or the columns of one [o2][o0*o1] matrix (order[2]) -- the same three cases
 * stock_fft_executor distinguishes with (stride, dist, blocks). */
class dfft_fft_executor {
public:
    template <typename index>
    dfft_fft_executor(box3d<index> const box, int dimension)
        : length(box.size[dimension]), o0(box.osize(0)), o1(box.osize(1)), o2(box.osize(2)),
          which(dimension == box.order[0] ? 0 : (dimension == box.order[1] ? 1 : 2)), total_size((long long)box.count()) {
        if (total_size > 0 && !dfft_length_supported(length))
            throw std::runtime_error("heffte dfft backend: FFT length " + std::to_string(length) +
                                     " is outside the single-pass range of libdfft_mi355x (7-smooth, <= 4096)");
    }
    ~dfft_fft_executor() {
        if (dev) (void)hipFree(dev);
    }
    dfft_fft_executor(dfft_fft_executor const&) = delete;
    dfft_fft_executor& operator=(dfft_fft_executor const&) = delete;

    void forward(std::complex<float> data[]) const { run(data, DFFT_F32, DFFT_FORWARD); }
    void backward(std::complex<float> data[]) const { run(data, DFFT_F32, DFFT_BACKWARD); }
    void forward(std::complex<double> data[]) const { run(data, DFFT_F64, DFFT_FORWARD); }
    void backward(std::complex<double> data[]) const { run(data, DFFT_F64, DFFT_BACKWARD); }

    /* real input: widen to complex, transform (heffte_backend_stock.h:475-496) */
    void forward(float const indata[], std::complex<float> outdata[]) const {
        for (long long i = 0; i < total_size; i++) outdata[i] = std::complex<float>(indata[i]);
        forward(outdata);
    }
    void backward(std::complex<float> indata[], float outdata[]) const {
        backward(indata);
        for (long long i = 0; i < total_size; i++) outdata[i] = std::real(indata[i]);
    }
    void forward(double const indata[], std::complex<double> outdata[]) const {
        for (long long i = 0; i < total_size; i++) outdata[i] = std::complex<double>(indata[i]);
        forward(outdata);
    }
    void backward(std::complex<double> indata[], double outdata[]) const {
        backward(indata);
        for (long long i = 0; i < total_size; i++) outdata[i] = std::real(indata[i]);
    }

    int box_size() const { return (int)total_size; }

private:
    template <typename T> void run(std::complex<T> data[], int dtype, int direction) const {
        if (total_size == 0) return;
        const size_t bytes = (size_t)total_size * sizeof(std::complex<T>);
        if (bytes > dev_bytes) {
            if (dev) dfft_shim::hip_ok(hipFree(dev), "hipFree");
            dev = nullptr;
            dfft_shim::hip_ok(hipMalloc(&dev, bytes), "hipMalloc");
            dev_bytes = bytes;
        }
        dfft_shim::hip_ok(hipMemcpy(dev, data, bytes, hipMemcpyHostToDevice), "copy to the device");
        int rc;
        if (which == 0) rc = dfft_fft1d_rows(dev, dev, length, (long long)o1 * o2, dtype, direction, nullptr);
        else if (which == 1) rc = dfft_fft1d_cols(dev, dev, length, o0, o2, dtype, direction, nullptr);
        else rc = dfft_fft1d_cols(dev, dev, length, (long long)o0 * o1, 1, dtype, direction, nullptr);
        dfft_shim::dfft_ok(rc, "1-D transform");
        dfft_shim::hip_ok(hipMemcpy(data, dev, bytes, hipMemcpyDeviceToHost), "copy from the device");  // synchronises
    }

    int            length, o0, o1, o2, which;
    long long      total_size;
    mutable void*  dev = nullptr;
    mutable size_t dev_bytes = 0;
};

/* Real-to-complex with shortening is not part of the slab pipeline this library accelerates. */
class dfft_fft_executor_r2c {
public:
    template <typename index>
    dfft_fft_executor_r2c(box3d<index> const box, int dimension) : rsize(box.count()), csize(box.r2c(dimension).count()) {}
    void forward(float const[], std::complex<float>[]) const { unsupported(); }
    void backward(std::complex<float> const[], float[]) const { unsupported(); }
    void forward(double const[], std::complex<double>[]) const { unsupported(); }
    void backward(std::complex<double> const[], double[]) const { unsupported(); }
    int  real_size() const { return rsize; }
    int  complex_size() const { return csize; }

private:
    static void unsupported() { throw std::runtime_error("heffte dfft backend: real-to-complex executors are not implemented (c2c only)"); }
    int rsize, csize;
};

template <> struct one_dim_backend<backend::stock> {
    using type = dfft_fft_executor;
    using type_r2c = dfft_fft_executor_r2c;

    template <typename index> static std::unique_ptr<dfft_fft_executor> make(void*, box3d<index> const box, int dimension) {
        return std::unique_ptr<dfft_fft_executor>(new dfft_fft_executor(box, dimension));
    }
    template <typename index> static std::unique_ptr<dfft_fft_executor> make(void*, box3d<index> const&, int, int) {
        throw std::runtime_error("2d dfft executor not implemented");
        return std::unique_ptr<dfft_fft_executor>();
    }
    template <typename index> static std::unique_ptr<dfft_fft_executor> make(void*, box3d<index> const&) {
        throw std::runtime_error("3d dfft executor not implemented");
        return std::unique_ptr<dfft_fft_executor>();
    }
    static bool can_merge2d() { return false; }
    static bool can_merge3d() { return false; }
    template <typename index> static std::unique_ptr<dfft_fft_executor_r2c> make_r2c(void*, box3d<index> const box, int dimension) {
        return std::unique_ptr<dfft_fft_executor_r2c>(new dfft_fft_executor_r2c(box, dimension));
    }
};

template <> struct default_plan_options<backend::stock> {
    static const bool use_reorder = true;  /* reshapes also reorder, so most lines arrive contiguous (rows) */
};

}  // namespace heffte

#endif /* HEFFTE_BACKEND_STOCK_FFT_H */
