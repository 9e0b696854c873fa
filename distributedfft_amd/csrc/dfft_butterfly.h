// dfft_butterfly.h -- in-register radix-2/3/4/5/8 DFT butterflies for the Stockham stages.
//
// Replaces (behaviour only) the butterfly bodies the reference's run-time code generator emits:
//   /root/reference/templateFFT/src/templateFFT.cpp:315-1075 (inlineRadixKernelFFT, cases 2/3/4/5/8).
// Everything here is static C++ for gfx950; there is no string code generator and no hiprtc.
//
// Conventions
//   V    = double2 (fp64 path) or float2 (fp32 path): one complex number, .x = re, .y = im; or cpair: the same element
//          of TWO adjacent fp32 columns, component-major (.x = both real parts, .y = both imaginary parts), so that every
//          butterfly line below compiles to packed-fp32 VALU instructions (v_pk_add/mul/fma_f32) and one lane moves 16 B.
//   DIR  = +1 forward  (kernel e^{-i theta}, reference FORWARD,  fft_mpi_common.h:18)
//        = -1 backward (kernel e^{+i theta}, reference BACKWARD, fft_mpi_common.h:19), unnormalised.
//   butterfly<R, DIR>(u): u[0..R-1] in, natural-order DFT of length R out (u[k] = sum_n u[n] w^{nk}).
#pragma once
#include <hip/hip_runtime.h>

namespace dfft {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct alignas(16) cpair {
    f32x2 x, y;
};

template <class V> struct real_of;
template <> struct real_of<double2> { using type = double; };
template <> struct real_of<float2>  { using type = float; };
template <> struct real_of<cpair>   { using type = float; };

template <class V> __device__ __forceinline__ V cadd(V a, V b) { return V{a.x + b.x, a.y + b.y}; }
template <class V> __device__ __forceinline__ V csub(V a, V b) { return V{a.x - b.x, a.y - b.y}; }
// data * twiddle: A is the data type (V), B a scalar complex (the twiddle type; for cpair both columns share it)
template <class A, class B> __device__ __forceinline__ A cmul(A a, B b) {
    // (a.x + i a.y)(b.x + i b.y); the compiler contracts these into fma chains.
    return A{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
// multiply by e^{-i DIR pi/2}: forward -> times (-i), backward -> times (+i)
template <int DIR, class V> __device__ __forceinline__ V mul_mi(V a) {
    if (DIR > 0) return V{a.y, -a.x};
    return V{-a.y, a.x};
}
template <class V, class R> __device__ __forceinline__ V cscale(V a, R s) { return V{a.x * s, a.y * s}; }

template <int R, int DIR, class V> struct Butterfly;

template <int DIR, class V> struct Butterfly<1, DIR, V> {
    static __device__ __forceinline__ void run(V*) {}
};

template <int DIR, class V> struct Butterfly<2, DIR, V> {
    static __device__ __forceinline__ void run(V* u) {
        V a = u[0], b = u[1];
        u[0] = cadd(a, b);
        u[1] = csub(a, b);
    }
};

template <int DIR, class V> struct Butterfly<3, DIR, V> {
    static __device__ __forceinline__ void run(V* u) {
        using Rt = typename real_of<V>::type;
        const Rt half = Rt(0.5), s32 = Rt(0.86602540378443864676372317075294);  // sin(pi/3)
        V t = cadd(u[1], u[2]);
        V d = cscale(csub(u[1], u[2]), s32);
        V m = V{u[0].x - half * t.x, u[0].y - half * t.y};
        V r = mul_mi<DIR>(d);
        u[0] = cadd(u[0], t);
        u[1] = cadd(m, r);
        u[2] = csub(m, r);
    }
};

template <int DIR, class V> struct Butterfly<4, DIR, V> {
    static __device__ __forceinline__ void run(V* u) {
        V t0 = cadd(u[0], u[2]), t1 = csub(u[0], u[2]);
        V t2 = cadd(u[1], u[3]), t3 = mul_mi<DIR>(csub(u[1], u[3]));
        u[0] = cadd(t0, t2);
        u[2] = csub(t0, t2);
        u[1] = cadd(t1, t3);
        u[3] = csub(t1, t3);
    }
};

template <int DIR, class V> struct Butterfly<5, DIR, V> {
    // Real parts in the Winograd form  m1,2 = (u0 - (t1 + t2)/4) +- (sqrt(5)/4)(t1 - t2)  [cos(2pi/5) = (sqrt5 - 1)/4,
    // cos(4pi/5) = -(sqrt5 + 1)/4]: a constant offset common to all five inputs cancels EXACTLY in (u0 - t5/4) and in
    // (t1 - t2), instead of surviving as offset * (1 + 2 c1 + 2 c2) = offset * O(eps).  The reference's batch tests
    // transform a ramp with values up to 2^26 (Test_1D.cpp:49-52); with the plain cosine form their round-trip error
    // is 1e-8, with this one 1e-13 like the published rows (templateFFT/csv/batch_result1D.csv:2-6).  Two real
    // multiplications per component instead of four.
    static __device__ __forceinline__ void run(V* u) {
        using Rt = typename real_of<V>::type;
        const Rt k5 = Rt(0.55901699437494742410229341718282);   // sqrt(5)/4
        const Rt s1 = Rt(0.95105651629515357211643933337938);   // sin(2pi/5)
        const Rt s2 = Rt(0.58778525229247312916870595463907);   // sin(4pi/5)
        V t1 = cadd(u[1], u[4]), t2 = cadd(u[2], u[3]);
        V t3 = csub(u[1], u[4]), t4 = csub(u[2], u[3]);
        V t5 = cadd(t1, t2);
        V m = V{u[0].x - Rt(0.25) * t5.x, u[0].y - Rt(0.25) * t5.y};
        V r = cscale(csub(t1, t2), k5);
        V m1 = cadd(m, r), m2 = csub(m, r);
        V q1 = mul_mi<DIR>(V{s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y});
        V q2 = mul_mi<DIR>(V{s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y});
        u[0] = cadd(u[0], t5);
        u[1] = cadd(m1, q1);
        u[4] = csub(m1, q1);
        u[2] = cadd(m2, q2);
        u[3] = csub(m2, q2);
    }
};

template <int DIR, class V> struct Butterfly<7, DIR, V> {
    static __device__ __forceinline__ void run(V* u) {
        using Rt = typename real_of<V>::type;
        // c_m = cos(2 pi m / 7), s_m = sin(2 pi m / 7)
        const Rt c1 = Rt(0.62348980185873353052500488400424), c2 = Rt(-0.22252093395631440428890256449679),
                 c3 = Rt(-0.90096886790241912623610231950745);
        const Rt s1 = Rt(0.78183148246802980870844452667406), s2 = Rt(0.97492791218182360701813168299393),
                 s3 = Rt(0.43388373911755812047576833284836);
        const V t1 = cadd(u[1], u[6]), t2 = cadd(u[2], u[5]), t3 = cadd(u[3], u[4]);
        const V d1 = csub(u[1], u[6]), d2 = csub(u[2], u[5]), d3 = csub(u[3], u[4]);
        const V a0 = u[0];
        // m_j = a0 + sum_k cos(2 pi j k / 7) t_k,  n_j = sum_k sin(2 pi j k / 7) d_k   (j k mod 7 folded to 1..3).
        // c1 + c2 + c3 = -1/2, so m_j = (a0 - t3/2) + c_{j1} (t1 - t3) + c_{j2} (t2 - t3): a constant offset common to all
        // seven inputs cancels exactly in each bracket (see Butterfly<5>)
        const V e1 = csub(t1, t3), e2 = csub(t2, t3);
        const V b0 = V{a0.x - Rt(0.5) * t3.x, a0.y - Rt(0.5) * t3.y};
        const V m1 = V{b0.x + c1 * e1.x + c2 * e2.x, b0.y + c1 * e1.y + c2 * e2.y};
        const V m2 = V{b0.x + c2 * e1.x + c3 * e2.x, b0.y + c2 * e1.y + c3 * e2.y};
        const V m3 = V{b0.x + c3 * e1.x + c1 * e2.x, b0.y + c3 * e1.y + c1 * e2.y};
        const V n1 = mul_mi<DIR>(V{s1 * d1.x + s2 * d2.x + s3 * d3.x, s1 * d1.y + s2 * d2.y + s3 * d3.y});
        const V n2 = mul_mi<DIR>(V{s2 * d1.x - s3 * d2.x - s1 * d3.x, s2 * d1.y - s3 * d2.y - s1 * d3.y});
        const V n3 = mul_mi<DIR>(V{s3 * d1.x - s1 * d2.x + s2 * d3.x, s3 * d1.y - s1 * d2.y + s2 * d3.y});
        u[0] = V{a0.x + t1.x + t2.x + t3.x, a0.y + t1.y + t2.y + t3.y};
        u[1] = cadd(m1, n1);
        u[6] = csub(m1, n1);
        u[2] = cadd(m2, n2);
        u[5] = csub(m2, n2);
        u[3] = cadd(m3, n3);
        u[4] = csub(m3, n3);
    }
};

template <int DIR, class V> struct Butterfly<8, DIR, V> {
    static __device__ __forceinline__ void run(V* u) {
        using Rt = typename real_of<V>::type;
        const Rt h = Rt(0.70710678118654752440084436210485);  // 1/sqrt(2)
        V e[4] = {u[0], u[2], u[4], u[6]};
        V o[4] = {u[1], u[3], u[5], u[7]};
        Butterfly<4, DIR, V>::run(e);
        Butterfly<4, DIR, V>::run(o);
        // o[k] *= W8^k, W8 = e^{-i DIR pi/4}
        V o1, o3;
        if (DIR > 0) {
            o1 = V{(o[1].x + o[1].y) * h, (o[1].y - o[1].x) * h};
            o3 = V{(o[3].y - o[3].x) * h, -(o[3].x + o[3].y) * h};
        } else {
            o1 = V{(o[1].x - o[1].y) * h, (o[1].x + o[1].y) * h};
            o3 = V{-(o[3].x + o[3].y) * h, (o[3].x - o[3].y) * h};
        }
        V o2 = mul_mi<DIR>(o[2]);
        u[0] = cadd(e[0], o[0]);
        u[4] = csub(e[0], o[0]);
        u[1] = cadd(e[1], o1);
        u[5] = csub(e[1], o1);
        u[2] = cadd(e[2], o2);
        u[6] = csub(e[2], o2);
        u[3] = cadd(e[3], o3);
        u[7] = csub(e[3], o3);
    }
};

}  // namespace dfft
