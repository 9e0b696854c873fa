/* slab_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C, fp64) of the reference's slab 3D C2C FFT.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's library; the product
 * (libdfft_mi355x.so) never links, calls or falls back to it.
 *
 * It restates, stage by stage and with the same buffers/index maps, what /root/reference/3dmpifft_opt does for P devices
 * emulated in one address space:
 *   input       fftSpeed3d_c2c.cpp:56-63      value(j) = j (re = im), j = global linear index
 *   slabs       fft_mpi_3d_api.cpp:84-133     xl = ceil(N0/P), yl = ceil(N1/P), last device takes the remainder
 *   t0 fftZY    fft_mpi_3d_api.cpp:466-522    per X-plane 2D FFT: Z (contiguous) then Y (stride N2)
 *   t1 pack     kernel_func.cpp:73-86         [xl][N1][N2] -> [d][xl][yl_d][N2], block d at d*x_size*yl*N2
 *   t2 exchange fft_mpi_3d_api.cpp:610-672    chunk(src->dst) lands at src*xl*yl_dst*N2 in dst's bufferDev1
 *   t3 fftX     fft_mpi_3d_api.cpp:524-573    transpose [N0][yl*N2] -> [yl*N2][N0] (kernels_201.cpp:56-57), 1D FFT of N0
 *   backward    fft_mpi_3d_api.cpp:203-212    the same stages in reverse order, e^{+i}, unnormalised
 * The 1D transform restates templateFFT's algorithm -- a Stockham autosort FFT with radix stages taken largest first
 * (templateFFT.cpp:4540-4588), per-stage twiddle w = e^{-+2 pi i (j mod S) / (S r)} (:337-341, LUT :5120-5141) and the
 * scatter pos = (j - j%S)*r + j%S + k*S (:1985-2045) -- with radices 8/4/2/3/5/7 and exact-trig twiddles.
 *
 * Pinning: tests/test_oracle_golden.py checks this file against heFFTe's golden vectors bundled with the reference
 * (test_units_stock.cpp:180-269, test_units_nompi.cpp:93-206) and against numpy.fft.fftn; on the GPU box the reference's
 * own GPU code (oracle/_ref) is run on the same input (tests/test_reference_parity.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct { double re, im; } cplx;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* radix plan: fold 2s into 8 then 4 then 2, then 3, 5, 7 (templateFFT.cpp:4540-4550); 0 on unsupported n */
static int make_plan(int n, int* radix) {
    int cnt = 0, m = n;
    while (m % 8 == 0) { radix[cnt++] = 8; m /= 8; }
    while (m % 4 == 0) { radix[cnt++] = 4; m /= 4; }
    while (m % 2 == 0) { radix[cnt++] = 2; m /= 2; }
    while (m % 3 == 0) { radix[cnt++] = 3; m /= 3; }
    while (m % 5 == 0) { radix[cnt++] = 5; m /= 5; }
    while (m % 7 == 0) { radix[cnt++] = 7; m /= 7; }
    /* oracle only: any other small prime factor as a by-definition DFT stage (lets the N = 11 golden vector through) */
    for (int p = 11; p <= 61 && m > 1; p += 2)
        while (m % p == 0) { radix[cnt++] = p; m /= p; }
    if (m != 1) return 0;
    return cnt;
}

/* One length-n Stockham FFT, in -> out via ping-pong (both length n, stride 1).  dir=+1: e^{-i}, dir=-1: e^{+i}. */
static void stockham(const cplx* in, cplx* out, cplx* tmp, int n, int dir, const int* radix, int nst, const cplx* wn) {
    /* wn[k] = e^{-2 pi i k / n}; conjugate on the fly for the inverse */
    const cplx* src = in;
    cplx* bufs[2];
    /* choose ping-pong so that the last stage writes `out` */
    bufs[(nst - 1) & 1] = out;
    bufs[nst & 1] = tmp;
    int S = 1;
    for (int st = 0; st < nst; ++st) {
        const int r = radix[st];
        const int nb = n / r; /* butterflies */
        cplx* dst = bufs[st & 1];
        for (int j = 0; j < nb; ++j) {
            cplx u[64], v[64];
            const int jm = j % S;
            for (int k = 0; k < r; ++k) {
                cplx x = src[j + k * nb];
                /* twiddle e^{-+2 pi i k jm / (S r)} = wn[k*jm*(n/(S*r))] */
                const cplx w = wn[(size_t)k * jm * (n / (S * r))];
                const double wi = dir > 0 ? w.im : -w.im;
                u[k].re = x.re * w.re - x.im * wi;
                u[k].im = x.re * wi + x.im * w.re;
            }
            /* radix-r DFT by definition */
            for (int q = 0; q < r; ++q) {
                double sr = 0, si = 0;
                for (int k = 0; k < r; ++k) {
                    const cplx w = wn[(size_t)((q * k) % r) * (n / r)];
                    const double wi = dir > 0 ? w.im : -w.im;
                    sr += u[k].re * w.re - u[k].im * wi;
                    si += u[k].re * wi + u[k].im * w.re;
                }
                v[q].re = sr;
                v[q].im = si;
            }
            const int base = (j - jm) * r + jm;
            for (int q = 0; q < r; ++q) dst[base + q * S] = v[q];
        }
        src = dst;
        S *= r;
    }
    if (nst == 0) out[0] = in[0];
}

typedef struct {
    int   n, nst, radix[40];
    cplx* wn;
    cplx *a, *b, *t; /* scratch lines */
} fft1d_t;

static int fft1d_init(fft1d_t* f, int n) {
    f->n = n;
    f->nst = make_plan(n, f->radix);
    if (n > 1 && f->nst == 0) return -1;
    f->wn = (cplx*)malloc(sizeof(cplx) * (size_t)n);
    f->a = (cplx*)malloc(sizeof(cplx) * (size_t)n);
    f->b = (cplx*)malloc(sizeof(cplx) * (size_t)n);
    f->t = (cplx*)malloc(sizeof(cplx) * (size_t)n);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int k = 0; k < n; ++k) {
        const long double a = two_pi * (long double)k / (long double)n;
        f->wn[k].re = (double)cosl(a);
        f->wn[k].im = (double)(-sinl(a));
    }
    return 0;
}
static void fft1d_free(fft1d_t* f) { free(f->wn); free(f->a); free(f->b); free(f->t); }

/* strided in-place line transform: data[i*stride], i < n */
static void fft1d_line(fft1d_t* f, cplx* data, long stride, int dir) {
    const int n = f->n;
    if (stride == 1) {
        stockham(data, f->b, f->t, n, dir, f->radix, f->nst, f->wn);
        memcpy(data, f->b, sizeof(cplx) * (size_t)n);
        return;
    }
    for (int i = 0; i < n; ++i) f->a[i] = data[(long)i * stride];
    stockham(f->a, f->b, f->t, n, dir, f->radix, f->nst, f->wn);
    for (int i = 0; i < n; ++i) data[(long)i * stride] = f->b[i];
}

static long slab_size(long n, int P, int g) { const long blk = (n + P - 1) / P; return g < P - 1 ? blk : n - (long)(P - 1) * blk; }

/* ---- exported -------------------------------------------------------------------------------------------------------- */

/* batched contiguous 1D C2C, in place: data[batch][n] interleaved doubles */
int oracle_fft1d(double* data, int n, long batch, int dir) {
    fft1d_t f;
    if (fft1d_init(&f, n)) return -1;
    for (long b = 0; b < batch; ++b) fft1d_line(&f, (cplx*)data + b * n, 1, dir);
    fft1d_free(&f);
    return 0;
}

/* driver input of device g (fftSpeed3d_c2c.cpp:56-63): out[j] = (first + j, first + j) */
void oracle_driver_input(double* out, long n0, long n1, long n2, int P, int g) {
    const long xl = (n0 + P - 1) / P, first = (long)g * xl * n1 * n2, cnt = slab_size(n0, P, g) * n1 * n2;
    for (long j = 0; j < cnt; ++j) out[2 * j] = out[2 * j + 1] = (double)(first + j);
}

/* Full slab pipeline over P emulated devices.
 *   forward : in  = [N0][N1][N2] (device g owns x in its slab), out = concatenation over d of [yl_d][N2][N0]
 *   backward: in  = concatenation over d of [yl_d][N2][N0],    out = [N0][N1][N2]; unnormalised.
 * stage_s[4] (optional) receives seconds spent in t0..t3 (backward: X, exchange, unpack, YZ) summed over devices. */
int oracle_slab_fft3d(const double* in_, double* out_, int n0, int n1, int n2, int P, int dir, double* stage_s) {
    const cplx* in = (const cplx*)in_;
    cplx*       out = (cplx*)out_;
    if (P < 1 || slab_size(n0, P, P - 1) < 1 || slab_size(n1, P, P - 1) < 1) return -2;
    fft1d_t fx, fy, fz;
    if (fft1d_init(&fx, n0) || fft1d_init(&fy, n1) || fft1d_init(&fz, n2)) return -1;
    const long xl = (n0 + P - 1) / P, yl = (n1 + P - 1) / P;
    double     ts[4] = {0, 0, 0, 0};
    /* per-device buffers: bufferDev1 / bufferDev2, each max(xs*N1*N2, N0*ys*N2) elements (getMaxDataCount :289-316) */
    cplx** b1 = (cplx**)malloc(sizeof(cplx*) * P);
    cplx** b2 = (cplx**)malloc(sizeof(cplx*) * P);
    for (int g = 0; g < P; ++g) {
        const long xs = slab_size(n0, P, g), ys = slab_size(n1, P, g);
        long       m = xs * n1 * n2, m2 = (long)n0 * ys * n2;
        if (m2 > m) m = m2;
        b1[g] = (cplx*)calloc((size_t)m, sizeof(cplx));
        b2[g] = (cplx*)calloc((size_t)m, sizeof(cplx));
    }
    if (dir > 0) {
        double t = now_s();
        for (int g = 0; g < P; ++g) { /* load + t0 */
            const long xs = slab_size(n0, P, g);
            memcpy(b1[g], in + (long)g * xl * n1 * n2, sizeof(cplx) * (size_t)(xs * n1 * n2));
            for (long xi = 0; xi < xs; ++xi) {
                cplx* plane = b1[g] + xi * n1 * n2;
                for (long y = 0; y < n1; ++y) fft1d_line(&fz, plane + y * n2, 1, dir);
                for (long z = 0; z < n2; ++z) fft1d_line(&fy, plane + z, n2, dir);
            }
        }
        ts[0] += now_s() - t; t = now_s();
        for (int g = 0; g < P; ++g) { /* t1 pack: kernel_func.cpp:73-86 */
            const long xs = slab_size(n0, P, g);
            for (long xi = 0; xi < xs; ++xi)
                for (long y = 0; y < n1; ++y) {
                    long d = y / yl; if (d > P - 1) d = P - 1;
                    const long yy = y - d * yl, yw = slab_size(n1, P, (int)d);
                    memcpy(b2[g] + d * xs * yl * n2 + (xi * yw + yy) * n2, b1[g] + (xi * n1 + y) * n2, sizeof(cplx) * (size_t)n2);
                }
        }
        ts[1] += now_s() - t; t = now_s();
        for (int g = 0; g < P; ++g) /* t2: chunk(g->d) -> b1[d] + g*xl*yl_d*N2 (:618-627) */
            for (int d = 0; d < P; ++d) {
                const long xs = slab_size(n0, P, g), yd = slab_size(n1, P, d);
                memcpy(b1[d] + (long)g * xl * yd * n2, b2[g] + (long)d * xs * yl * n2, sizeof(cplx) * (size_t)(xs * yd * n2));
            }
        ts[2] += now_s() - t; t = now_s();
        long ooff = 0;
        for (int d = 0; d < P; ++d) { /* t3: transpose 201 then FFT along X */
            const long ys = slab_size(n1, P, d), rows = ys * n2;
            for (long x = 0; x < n0; ++x)
                for (long c = 0; c < rows; ++c) b2[d][c * n0 + x] = b1[d][x * rows + c];
            for (long c = 0; c < rows; ++c) fft1d_line(&fx, b2[d] + c * n0, 1, dir);
            memcpy(out + ooff, b2[d], sizeof(cplx) * (size_t)(rows * n0));
            ooff += rows * n0;
        }
        ts[3] += now_s() - t;
    } else {
        double t = now_s();
        long   ioff = 0;
        for (int d = 0; d < P; ++d) { /* inverse X FFT then transpose 120 -> [x][yl_d][N2] */
            const long ys = slab_size(n1, P, d), rows = ys * n2;
            memcpy(b1[d], in + ioff, sizeof(cplx) * (size_t)(rows * n0));
            ioff += rows * n0;
            for (long c = 0; c < rows; ++c) fft1d_line(&fx, b1[d] + c * n0, 1, dir);
            for (long x = 0; x < n0; ++x)
                for (long c = 0; c < rows; ++c) b2[d][x * rows + c] = b1[d][c * n0 + x];
        }
        ts[0] += now_s() - t; t = now_s();
        for (int d = 0; d < P; ++d) /* exchange back */
            for (int g = 0; g < P; ++g) {
                const long xs = slab_size(n0, P, g), yd = slab_size(n1, P, d);
                memcpy(b1[g] + (long)d * xs * yl * n2, b2[d] + (long)g * xl * yd * n2, sizeof(cplx) * (size_t)(xs * yd * n2));
            }
        ts[1] += now_s() - t; t = now_s();
        for (int g = 0; g < P; ++g) { /* unpack: kernel_func.cpp:88-100 */
            const long xs = slab_size(n0, P, g);
            for (long xi = 0; xi < xs; ++xi)
                for (long y = 0; y < n1; ++y) {
                    long d = y / yl; if (d > P - 1) d = P - 1;
                    const long yy = y - d * yl, yw = slab_size(n1, P, (int)d);
                    memcpy(b2[g] + (xi * n1 + y) * n2, b1[g] + d * xs * yl * n2 + (xi * yw + yy) * n2, sizeof(cplx) * (size_t)n2);
                }
        }
        ts[2] += now_s() - t; t = now_s();
        for (int g = 0; g < P; ++g) { /* inverse Y then Z per plane */
            const long xs = slab_size(n0, P, g);
            for (long xi = 0; xi < xs; ++xi) {
                cplx* plane = b2[g] + xi * n1 * n2;
                for (long z = 0; z < n2; ++z) fft1d_line(&fy, plane + z, n2, dir);
                for (long y = 0; y < n1; ++y) fft1d_line(&fz, plane + y * n2, 1, dir);
            }
            memcpy(out + (long)g * xl * n1 * n2, b2[g], sizeof(cplx) * (size_t)(xs * n1 * n2));
        }
        ts[3] += now_s() - t;
    }
    if (stage_s) memcpy(stage_s, ts, sizeof(ts));
    for (int g = 0; g < P; ++g) { free(b1[g]); free(b2[g]); }
    free(b1); free(b2);
    fft1d_free(&fx); fft1d_free(&fy); fft1d_free(&fz);
    return 0;
}
