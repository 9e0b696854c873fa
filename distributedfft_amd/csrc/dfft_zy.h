// dfft_zy.h -- the one-launch form of t0 (dfft_zy.hip).  Internal header.
#pragma once
#include <hip/hip_runtime.h>

namespace dfft {

enum { ZY_MAX_PLANES = 4096 };

// control block in device memory, zeroed once when the plan is created (launches count on from where the last one stopped)
struct alignas(128) ZyCtl {
    unsigned ticket;               // next work item
    unsigned pad0[31];
    unsigned error;                // != 0: a consumer unit waited longer than 20 ms for its plane (the launch gave up)
    unsigned pad1[31];
    unsigned done[ZY_MAX_PLANES];  // per plane: producer units that have published their results
};

struct ZyLaunch {
    int         dtype;      // DType (F64 only)
    int         n1, n2;     // Y and Z lengths
    int         dir;        // +1: Z rows src -> w, then Y columns in place on w;  -1: Y columns in place on w, then Z rows w -> dst
    const void* src;        // forward: [plane][N1][N2], planes src_plane elements apart
    void*       w;          // hand-over buffer: rows N2 apart, planes w_plane elements apart
    void*       dst;        // backward: [plane][N1][N2], planes dst_plane elements apart
    long long   src_plane, w_plane, dst_plane;
    long long   nplanes, chunk;  // planes; planes per Infinity-Cache phase
    ZyCtl*      ctl;
    unsigned    generation;  // how many launches this control block has served (all with the same geometry and direction)
    const void *twz, *twy;  // N2- and N1-entry twiddle tables (fp64)
};

bool       zy_supported(int dtype, int n1, int n2);
hipError_t launch_zy(const ZyLaunch& L, hipStream_t stream);

}  // namespace dfft
