// dfft_fft_inst.hip -- explicit instantiations of the FFT kernel template, one group of lengths per
// translation unit (compile with -DDFFT_INST_GROUP=<g>, g in [0, DFFT_NUM_INST_GROUPS)).
#include "dfft_fft_impl.h"
#include "dfft_plans.h"

#ifndef DFFT_INST_GROUP
#error "compile with -DDFFT_INST_GROUP=<g>"
#endif

namespace dfft {

template <int N> struct PlanFor;
#define DFFT_DECL_PLAN(N, GRP, E, ...) \
    template <> struct PlanFor<N> { using type = Plan<N, E, __VA_ARGS__>; static constexpr int group = GRP; };
DFFT_PLAN_TABLE(DFFT_DECL_PLAN)
#undef DFFT_DECL_PLAN

// fp32 uses the table's plan unless a length is listed here (measured: 16 points/thread helps 1024-point fp64 columns,
// 3.4 -> 4.4 TB/s, but costs fp32 -- 16-column tiles then need 1024-thread blocks: 1024^3 fp32 7.4 -> 12.4 ms in t0).
template <int N> struct PlanFor32 : PlanFor<N> {};
// (512-point fp32 with 16 points per thread was also tried: X pass 0.567 -> 0.524 ms but the Z+Y stage 0.70 -> 1.00 ms.)
template <> struct PlanFor32<1024> { using type = Plan<1024, 8, 8, 8, 8, 2>; };

// Scalar float2 COLUMN fall-back (launch_scalar32: fp32 column launches that cannot run on column pairs): workgroups of at most 512
// threads, so 1024 points run on the table's 16 points x 64 threads per column (8-column tiles, 64-byte pieces) instead of the fp32
// plan's 8 x 128.  (2048 points on the row plan's 32 x 64 were tried for the same reason: 268-388 bytes of scratch; they keep 16 x 128
// on 4-column tiles.)
template <int N> struct PlanScalar32 : PlanFor32<N> {};
template <> struct PlanScalar32<1024> : PlanFor<1024> {};

// Row launches may use a plan of their own (launch_rows): 2048 contiguous points as one wave64 with 32 points per thread
// (no s_barrier: 4.5 -> 5.3 TB/s fp64, 4.1 -> 5.1 fp32); the column kernel stays at 16 points per thread.
template <int N> struct PlanRows : PlanFor<N> {};
template <int N> struct PlanRows32 : PlanFor32<N> {};
template <> struct PlanRows<2048> { using type = Plan<2048, 32, 8, 8, 8, 4>; };
template <> struct PlanRows32<2048> { using type = Plan<2048, 32, 8, 8, 8, 4>; };
// 768 contiguous points: 12 points x 64 threads -- one wavefront per row, radix 4 4 4 4 3 -- instead of the table's 24 x 32 (round 4,
// profiles/r04/experiments/lib_ab_768_e12.log, rows_768_e12.log: t0 of 512 x 512 x 768 fp64 2.26 -> 2.17 ms, 768^3 5.54 -> 5.26; as a column
// length the 24-point plan stays -- config 4's Y axis measured equal)
template <> struct PlanRows<768> { using type = Plan<768, 12, 4, 4, 4, 4, 3>; };
template <> struct PlanRows32<768> { using type = Plan<768, 12, 4, 4, 4, 4, 3>; };
// 3125 contiguous points: 625 threads x 5 points instead of 125 x 25 (two workgroups = 20 waves per CU instead of 4)
template <> struct PlanRows<3125> { using type = Plan<3125, 5, 5, 5, 5, 5, 5>; };
template <> struct PlanRows32<3125> { using type = Plan<3125, 5, 5, 5, 5, 5, 5>; };

// Half plan for the DIF-split full-line column tiles (launch_plan): 2048-point columns run as two 1024-point transforms
template <int N> struct PlanHalf { using type = void; };
template <> struct PlanHalf<2048> { using type = PlanFor<1024>::type; };
template <> struct PlanHalf<1024> { using type = PlanFor<512>::type; };  // used from DFFT_DIF2_MIN=1024 on (measurement switch)

template <int N> hipError_t launch_n(const FftLaunch& L, hipStream_t stream) {
    if (!L.cols) {
        if (L.dtype == F64) return launch_rows<double2, typename PlanRows<N>::type>(L, stream);
        if (L.dtype == F32) return launch_rows<float2, typename PlanRows32<N>::type>(L, stream);
        return hipErrorInvalidValue;
    }
    if (L.dtype == F64) return launch_plan<double2, typename PlanFor<N>::type, typename PlanHalf<N>::type>(L, stream);
    if (L.dtype == F32) {
        // column launches on even column counts run on column pairs with the fp64 geometry (16 bytes per lane)
        FftLaunch Lp;
        if (make_pair_launch<typename PlanFor<N>::type>(L, Lp)) return launch_plan<cpair, typename PlanFor<N>::type, typename PlanHalf<N>::type>(Lp, stream);
        return launch_scalar32<typename PlanScalar32<N>::type>(L, stream);
    }
    return hipErrorInvalidValue;
}

#define DFFT_INST_PLAN(N, GRP, E, ...) DFFT_INST_IF_##GRP(N)
#define DFFT_DO_INST(N) template hipError_t launch_n<N>(const FftLaunch&, hipStream_t);

#if DFFT_INST_GROUP == 0
#define DFFT_INST_IF_0(N) DFFT_DO_INST(N)
#else
#define DFFT_INST_IF_0(N)
#endif
#if DFFT_INST_GROUP == 1
#define DFFT_INST_IF_1(N) DFFT_DO_INST(N)
#else
#define DFFT_INST_IF_1(N)
#endif
#if DFFT_INST_GROUP == 2
#define DFFT_INST_IF_2(N) DFFT_DO_INST(N)
#else
#define DFFT_INST_IF_2(N)
#endif
#if DFFT_INST_GROUP == 3
#define DFFT_INST_IF_3(N) DFFT_DO_INST(N)
#else
#define DFFT_INST_IF_3(N)
#endif
#if DFFT_INST_GROUP == 4
#define DFFT_INST_IF_4(N) DFFT_DO_INST(N)
#else
#define DFFT_INST_IF_4(N)
#endif
#if DFFT_INST_GROUP == 5
#define DFFT_INST_IF_5(N) DFFT_DO_INST(N)
#else
#define DFFT_INST_IF_5(N)
#endif
#if DFFT_INST_GROUP == 6
#define DFFT_INST_IF_6(N) DFFT_DO_INST(N)
#else
#define DFFT_INST_IF_6(N)
#endif
#if DFFT_INST_GROUP == 7
#define DFFT_INST_IF_7(N) DFFT_DO_INST(N)
#else
#define DFFT_INST_IF_7(N)
#endif
#if DFFT_INST_GROUP == 8
#define DFFT_INST_IF_8(N) DFFT_DO_INST(N)
#else
#define DFFT_INST_IF_8(N)
#endif
#if DFFT_INST_GROUP == 9
#define DFFT_INST_IF_9(N) DFFT_DO_INST(N)
#else
#define DFFT_INST_IF_9(N)
#endif
#if DFFT_INST_GROUP == 10
#define DFFT_INST_IF_10(N) DFFT_DO_INST(N)
#else
#define DFFT_INST_IF_10(N)
#endif
#if DFFT_INST_GROUP == 11
#define DFFT_INST_IF_11(N) DFFT_DO_INST(N)
#else
#define DFFT_INST_IF_11(N)
#endif

DFFT_PLAN_TABLE(DFFT_INST_PLAN)

}  // namespace dfft
