// dfft_plans.h -- compile-time radix plans (N, points per thread E, radix sequence), largest radix first.
//
// Takes over the *decisions* of the reference's run-time scheduler (templateFFT.cpp:3941-4607 FFTScheduler:
// fold 2s into radix 8 then 4, largest radix first, 8 complex registers per thread for powers of two,
// 12/24 when a factor 3 is present) as a static table; nothing is generated at run time.
// X(N, group, E, radices...) -- "group" only spreads the instantiations over translation units.
// Powers of 3 and 5 (the reference's published component rows 243, 625, 2187, 3125: templateFFT/csv/batch_result1D.csv:5,6,14,16;
// 2D 243^2, 729x243: batch_result2D.csv:32,33) run radix-3 / radix-5 stages with 9 (25 for 3125) points per thread.
// 4096 and 2401 = 7^4 close the reference's single-pass range (templateFFT.cpp:3946: every power of two up to 4096 in one
// upload); contiguous rows of 4096 and 3125 points use plans of their own (dfft_fft_inst.hip: 8 and 5 points per thread).
#pragma once

// 768 points (config 4's Y axis): 12 points x 64 threads per column (radix 4 4 4 4 3) -- a 512-thread workgroup per 8-column tile, the
// geometry of every other plane shape of the one-launch YZ stage (dfft_zy.hip), which is what this plan is chosen for: as a two-launch
// column kernel it measures equal to 24 points x 32 threads (radix 8 8 4 3, 256-thread workgroups; profiles/r04/experiments/
// lib_ab_768_e12.log, profiles/r05/experiments/variant_ab_768_e12.log), inside the one-launch stage 24 x 32 leaves four waves per CU for
// the row units and loses 20 % (profiles/r05/README.md section 3).  -DDFFT_768_E12=0 builds the 24-point plan.
#ifndef DFFT_768_E12
#define DFFT_768_E12 1
#endif
#if DFFT_768_E12
#define DFFT_PLAN_768(X) X(768, 4, 12, 4, 4, 4, 4, 3)
#else
#define DFFT_PLAN_768(X) X(768, 4, 24, 8, 8, 4, 3)
#endif

#define DFFT_PLAN_TABLE(X)        \
    X(2, 0, 2, 2)                 \
    X(3, 0, 3, 3)                 \
    X(4, 0, 4, 4)                 \
    X(5, 0, 5, 5)                 \
    X(6, 0, 6, 3, 2)              \
    X(7, 0, 7, 7)                 \
    X(8, 0, 8, 8)                 \
    X(9, 0, 3, 3, 3)              \
    X(10, 0, 10, 5, 2)            \
    X(12, 0, 12, 4, 3)            \
    X(14, 0, 14, 7, 2)            \
    X(16, 0, 4, 4, 4)             \
    X(24, 0, 24, 8, 3)            \
    X(25, 0, 5, 5, 5)             \
    X(32, 1, 8, 8, 4)             \
    X(48, 1, 24, 8, 3, 2)         \
    X(49, 1, 7, 7, 7)             \
    X(64, 1, 8, 8, 8)             \
    X(96, 1, 24, 8, 4, 3)         \
    X(100, 1, 20, 5, 5, 4)        \
    X(125, 1, 5, 5, 5, 5)         \
    X(128, 2, 8, 8, 8, 2)         \
    X(192, 2, 24, 8, 8, 3)        \
    X(256, 2, 8, 8, 8, 4)         \
    X(343, 2, 7, 7, 7, 7)         \
    X(384, 3, 24, 8, 8, 3, 2)     \
    X(512, 3, 8, 8, 8, 8)         \
    DFFT_PLAN_768(X)              \
    X(1024, 5, 16, 8, 8, 8, 2)    \
    X(2048, 6, 16, 8, 8, 8, 4)    \
    X(40, 7, 20, 5, 4, 2)         \
    X(80, 7, 20, 5, 4, 4)         \
    X(160, 7, 20, 5, 4, 4, 2)     \
    X(200, 7, 20, 5, 5, 4, 2)     \
    X(320, 7, 20, 5, 4, 4, 4)     \
    X(400, 8, 20, 5, 5, 4, 4)     \
    X(640, 8, 20, 5, 4, 4, 4, 2)  \
    X(1000, 8, 10, 5, 5, 5, 2, 2, 2) \
    X(1280, 9, 20, 5, 4, 4, 4, 4) \
    X(1536, 9, 24, 8, 8, 8, 3)    \
    X(27, 0, 9, 3, 3, 3)          \
    X(81, 1, 9, 3, 3, 3, 3)       \
    X(243, 2, 9, 3, 3, 3, 3, 3)   \
    X(729, 4, 9, 3, 3, 3, 3, 3, 3) \
    X(2187, 6, 9, 3, 3, 3, 3, 3, 3, 3) \
    X(625, 5, 5, 5, 5, 5, 5)      \
    X(3125, 3, 25, 5, 5, 5, 5, 5) \
    X(4096, 10, 16, 8, 8, 8, 8)   \
    X(2401, 11, 7, 7, 7, 7, 7)

#define DFFT_NUM_INST_GROUPS 12
