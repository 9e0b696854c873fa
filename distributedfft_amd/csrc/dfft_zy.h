// dfft_zy.h -- the one-launch form of t0 (dfft_zy.hip).  Internal header.
#pragma once
#include <hip/hip_runtime.h>

#include "dfft_kernels.h"

namespace dfft {

enum { ZY_MAX_PLANES = 4096 };

// -DDFFT_ZY_ROW_PITCH=1 (build-time experiment of the end of round 4, measured and NOT adopted): the row pitch of the hand-over buffer w
// becomes a launch parameter instead of the compile-time N2, so that a plan whose buffer has padded rows (DFFT_PAD_ROW) can use the
// one-launch stage.  Meant for BACKWARD single-GPU plans, whose column units read w from HBM at a power-of-two row stride (HISTORY.md
// section 8-7).  Same digests as the shipped build, no gain: inverse YZ stage of 512^3 fp64 1.42 ms with natural rows, 1.45 with padded
// ones (profiles/r04/experiments/backward_row_pitch.log).  The default build is byte-identical without it (library d7893003).
#ifndef DFFT_ZY_ROW_PITCH
#define DFFT_ZY_ROW_PITCH 0
#endif
#if DFFT_ZY_ROW_PITCH
#define DFFT_ZY_SET_PITCH(L, v) (L).w_pitch = (v)
#else
#define DFFT_ZY_SET_PITCH(L, v) ((void)0)  // (keeps dfft_plan.cpp's line numbers -- they are part of its error strings -- and object code as profiled)
#endif

// why a launch gave up (ZyCtl::error and the host-visible word)
enum { ZY_ERR_TIMEOUT = 1,   // a consumer unit polled spin_polls times for its plane's producers
       ZY_ERR_DESYNC = 2 };  // the first ticket of a launch did not fit its ticket_base: counters out of step with the host

// control block in device memory, zeroed once when the plan is created (launches count on from where the last one stopped)
struct alignas(128) ZyCtl {
    unsigned ticket;               // next work item
    unsigned pad0[31];
    unsigned error;                // != 0 (ZY_ERR_*): a launch gave up; sticky -- every later launch on this block returns at once
    unsigned pad1[31];
    unsigned done[ZY_MAX_PLANES];  // per plane: producer units that have published their results
};

struct ZyLaunch {
    int         dtype;      // DType (F64 only)
    int         n1, n2;     // Y and Z lengths
    int         dir;        // +1: Z rows src -> w, then Y columns in place on w;  -1: Y columns in place on w, then Z rows w -> dst
    int         sign;       // direction of the transform when it is not `dir` (0 = dir): -1 with dir = +1 runs the INVERSE stage rows first (lazy, un-packed)
    const void* src;        // forward: [plane][N1][N2], planes src_plane elements apart
    void*       w;          // hand-over buffer: rows N2 apart, planes w_plane elements apart
    void*       dst;        // backward: [plane][N1][N2], planes dst_plane elements apart
    long long   src_plane, w_plane, dst_plane;
#if DFFT_ZY_ROW_PITCH
    long long   w_pitch;      // distance between consecutive rows of w (elements; >= n2)
#endif
    long long   plane0, nplanes, chunk;  // first plane and number of planes of this launch; planes per Infinity-Cache phase
    ZyCtl*      ctl;
    unsigned    ticket_base;  // value of ctl->ticket when this launch starts (the host adds zy_tickets() per launch)
    unsigned    done_base;    // value of ctl->done[plane] of this launch's planes when it starts (producers per plane x executes so far)
    int         packed;       // the column side that is not w is the packed exchange layout `pk` (forward: dst, backward: src)
    AxisMap     pk;           // FFT index -> offset inside the packed layout (blocks of pk.blk rows per destination / sub-block)
    long long   pk_plane;     // distance between consecutive X planes inside a block of the packed layout
    RotMap      rot;          // rot != 0: rows of the packed layout are rotated by rot * (plane + a0) elements (mask = N2 - 1)
    const void *twz, *twy;  // N2- and N1-entry twiddle tables (fp64)
    int         lazy;         // the lazy-publish variant of the kernel (the default; DFFT_ZY_LAZY=0: eager)
    int         fault;        // test hook: the consumers of this launch wait for one producer more than a plane has
    unsigned*   err_host;     // device pointer of a pinned host word: written with ZY_ERR_* when the launch gives up
    unsigned    spin_polls;   // bound of a consumer's wait, in polls of its plane's counter (1-3 us each)
    unsigned*   part_done;    // != nullptr (forward packed lazy launches): per-part counters of finished column units, part = (plane - plane0) / part_planes
    long long   part_planes;
};

bool       zy_supported(int dtype, int n1, int n2);
int        zy_col_threads(int n1, int lazy);
long long  zy_grid();
unsigned   zy_units_per_plane(int n1, int n2, int dir, int packed, unsigned* producers);
unsigned   zy_tickets(int n1, int n2, int dir, int packed, long long nplanes, long long chunk);
hipError_t launch_zy(const ZyLaunch& L, hipStream_t stream);
// exchange stream: wait until *ctr has reached target (the column units of one X-plane part of a part_done launch)
hipError_t launch_zy_part_wait(const unsigned* ctr, unsigned target, const ZyCtl* ctl, unsigned* err_host, hipStream_t stream);

}  // namespace dfft
