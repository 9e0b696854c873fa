// dfft_plan.cpp -- slab bookkeeping, plan object and the t0..t3 execute sequence behind the C-ABI (include/dfft.h).
//
// Reference being replaced (behaviour, not code): /root/reference/3dmpifft_opt/include/fft_mpi_3d_api.cpp
//   fft_mpi_init :3-39, fft_mpi_plan_dft_c2c_3d :41-141, fft_mpi_destroy_plan :143-179,
//   fft_mpi_execute_dft_3d_c2c :181-214, getProperDeviceNum :232-272, getDataCountForNode :274-287,
//   getMaxDataCount :289-316, fftZY :466-522, fftX :524-573, localTransposeUneven :575-608, slabAlltoall :610-672.
//
// MI355X-first differences (DESIGN.md has the full list):
//   * t0 is two launches over the WHOLE slab (Z rows, Y columns), not 2*xl per-plane launches;
//   * fused mode: the Y pass stores straight into the packed [dest][xl][yl][N2] send layout (t1 disappears) and the
//     X pass loads [x][yl][N2] column tiles and stores [yl][N2][kx] (the 16x16 tile transpose disappears):
//     6*S bytes per element of HBM traffic for the local pipeline instead of the reference's 10*S;
//   * P == 1 short-circuits the exchange (the reference does a full-size self copy);
//   * everything is enqueued on one HIP stream, stage boundaries are HIP events; host-blocking per-stage timing is
//     opt-in (DFFT_EXEC_SYNC_STAGES) for drop-in comparability with the reference's MPI_Wtime brackets.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <numeric>
#include <string>
#include <tuple>
#include <thread>

#include "dfft_internal.h"
#include "dfft_long.h"
#include "dfft_zy.h"

namespace dfft {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
int  fail(int code, const std::string& msg) {
    g_last_error = msg;
    trace_on_error(code, msg);  // communication failures print the process's last control-plane events (dfft_trace.cpp)
    return code;
}

// ---------------------------------------------------------------------------------------------------------------
// twiddle tables
struct TwKey {
    int  dev, n, dtype;
    bool operator<(const TwKey& o) const {
        if (dev != o.dev) return dev < o.dev;
        if (n != o.n) return n < o.n;
        return dtype < o.dtype;
    }
};
static std::mutex               g_tw_mutex;
static std::map<TwKey, void*>   g_tw_cache;

int get_twiddles(int n, int dtype, const void** table) {
    int dev = 0;
    DFFT_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_tw_mutex);
    TwKey key{dev, n, dtype};
    auto it = g_tw_cache.find(key);
    if (it != g_tw_cache.end()) {
        *table = it->second;
        return DFFT_OK;
    }
    // e^{-2 pi i k / n} evaluated in extended precision and rounded once (the reference builds its LUT with host
    // double cos/sin, templateFFT.cpp:5120-5141).
    const long double two_pi = 6.283185307179586476925286766559005768L;
    void*             dptr = nullptr;
    if (dtype == DFFT_F64) {
        std::vector<double> h(2 * (size_t)n);
        for (int k = 0; k < n; ++k) {
            const long double a = two_pi * (long double)k / (long double)n;
            h[2 * k] = (double)cosl(a);
            h[2 * k + 1] = (double)(-sinl(a));
        }
        DFFT_HIP_TRY(hipMalloc(&dptr, h.size() * sizeof(double)));
        DFFT_HIP_TRY(hipMemcpy(dptr, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    } else {
        std::vector<float> h(2 * (size_t)n);
        for (int k = 0; k < n; ++k) {
            const long double a = two_pi * (long double)k / (long double)n;
            h[2 * k] = (float)cosl(a);
            h[2 * k + 1] = (float)(-sinl(a));
        }
        DFFT_HIP_TRY(hipMalloc(&dptr, h.size() * sizeof(float)));
        DFFT_HIP_TRY(hipMemcpy(dptr, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    g_tw_cache[key] = dptr;
    *table = dptr;
    return DFFT_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// launch helpers: the three passes expressed as address maps of the one FFT kernel template
static AxisMap plain_axis(long long n, long long stride, long long cstride) {
    AxisMap m;
    m.blk = (int)n;
    m.nblk = 1;
    m.blk_stride = 0;
    m.stride = stride;
    m.cstride = cstride;
    m.last_delta = 0;
    m.sub = 1;
    m.sub_stride = 0;
    return m;
}

// experiment knobs: DFFT_Z_GRID / DFFT_Y_GRID / DFFT_X_GRID cap the persistent grid of the respective pass -- a tuning knob and the
// tests' way to give a workgroup many tiles.  Read ONCE per plan (dfft_plan_create), never on the launch path: getenv is not safe
// against a concurrent setenv of another device thread.
static int env_grid(const char* name) {
    const char* e = getenv(name);
    return e ? atoi(e) : 0;
}

static int check_launch(hipError_t e, const char* what) {
    if (e == hipSuccess) return DFFT_OK;
    if (e == hipErrorInvalidValue) return fail(DFFT_EUNSUPPORTED, std::string(what) + ": no gfx950 kernel for this length/precision");
    return fail(DFFT_EHIP, std::string(what) + ": " + hipGetErrorString(e));
}

// Where the rows of an [x][y][z] slab sit: elements between consecutive rows and between consecutive planes.  The natural
// layout is {N2, N1*N2}; the plan's padded work buffer (dfft_plan_s::wbuf) uses {N2 + one line, N1*pitch + one line}.
struct SlabLayout {
    long long pitch, plane;
};

// scratch slab of the plan that is executing on this thread (long-axis plans; set by dfft_execute)
static thread_local void* t_plan_scratch = nullptr;
static thread_local int   t_plan_zgrid = 0;  // the executing plan's cap on its row launches' grids (dfft_plan_s::grid_z)

// contiguous rows: `rows` FFTs of length n; row pitch n, or (lin/lout given) rows_per_plane rows per plane in the given layouts
static int fft_rows(const void* in, void* out, int n, long long rows, int dtype, int dir, hipStream_t s,
                    long long first_row = 0, int hints = 0, double scale = 1.0, const SlabLayout* lin = nullptr,
                    const SlabLayout* lout = nullptr, long long rows_per_plane = 0, void* long_scratch_buf = nullptr, int grid_limit = 0) {
    if (n > 4096) {  // beyond the single-pass range: four-step decomposition (dfft_long.hip), plain contiguous rows only
        if (lin || lout) return fail(DFFT_EINVAL, "fft_rows: long axes use the natural layout");
        if (rows <= 0) return DFFT_OK;
        const size_t off = (size_t)first_row * n * elem_bytes(dtype);
        void*        scr = long_scratch_buf ? long_scratch_buf : t_plan_scratch;
        const bool   own = !scr;
        LongScratchLease lease = nullptr;
        if (own) scr = long_scratch((size_t)rows * n * elem_bytes(dtype), s, &lease);
        if (!scr) return fail(DFFT_EHIP, "fft_rows: cannot allocate the scratch buffer of the four-step transform");
        const int rc = long_fft((const char*)in + off, (char*)out + off, n, 1, rows, dtype, dir, scale, scr, s);
        long_scratch_release(lease);
        return rc;
    }
    const void* tw = nullptr;
    int         rc = get_twiddles(n, dtype, &tw);
    if (rc) return rc;
    FftLaunch L;
    std::memset(&L, 0, sizeof(L));
    L.dtype = dtype;
    L.n = n;
    L.dir = dir;
    L.cols = 0;
    L.in = in;
    L.out = out;
    L.tw = tw;
    L.imap = L.omap = plain_axis(n, 1, 0);
    L.itile = L.otile = TileMap{(long long)n, 0};
    L.ntiles = rows;
    L.a_first = first_row;
    L.hints = hints;
    L.scale = scale;
    L.tiles_per_a = 1;
    L.ncols = 1;
    if (lin && lout && rows_per_plane > 0) {  // tile = (plane a, row b): base = a * plane + b * pitch (CB = 1 for rows)
        if (first_row % rows_per_plane != 0 || rows_per_plane >= (1ll << 31)) return fail(DFFT_EINVAL, "fft_rows: chunk is not whole planes");
        L.itile = TileMap{lin->plane, lin->pitch};
        L.otile = TileMap{lout->plane, lout->pitch};
        L.tiles_per_a = (int)rows_per_plane;
        L.a_first = first_row / rows_per_plane;
    }
    L.grid_limit = grid_limit ? grid_limit : t_plan_zgrid;  // (dfft_plan_s::grid_z: DFFT_Z_GRID when the plan was created)
    return check_launch(launch_fft(L, s), "fft_rows");
}

}  // namespace dfft

using namespace dfft;

// ---------------------------------------------------------------------------------------------------------------
struct dfft_plan_s {
    long long   N[3];
    int         dtype, direction;
    int         P, me;
    unsigned    flags;
    bool        inplace, is_last;
    double      scale = 1.0;  // folded into the X pass (dfft_plan_set_scale)
    bool        exch = false;  // t2 runs: P > 1, or DFFT_FORCE_EXCHANGE=1 with an RCCL communicator (single-GPU tests)
    long long   max_count;
    Slab        sx, sy;       // X slabs (before), Y slabs (after)
    long long   xs, ys;       // this device's extents
    void *      in, *out, *buf1, *buf2;
    dfft_comm_t comm;
    int         device;
    hipStream_t stream;
    hipEvent_t  ev[6];  // [0..4] stage boundaries, [5] between the two FFT kernels of the YZ stage
    double      host_t[4];
    bool        host_timed;
    bool        timed = true;  // the last execute recorded its stage events (false: DFFT_EXEC_NO_TIMING)
    ExchangeDesc xd;
    ExchangeDesc xd2;        // DFFT_PLAN_NATURAL: the second (Y -> X) exchange
    long long   chunk_planes;  // planes per Z+Y chunk (Infinity-Cache blocking); 0 = whole slab in one launch pair
    // DFFT_PLAN_OVERLAP (forward, P > 1): exchange parts on a second stream behind the plane-chunked Z+Y passes
    long long               part_planes = 0;  // planes per exchange part, identical on every rank; 0 = overlap off
    hipStream_t             stream2 = nullptr;
    void*                   rbuf = nullptr;  // dedicated receive buffer of the overlapped exchange
    hipEvent_t              join_ev = nullptr;
    std::vector<hipEvent_t> part_ev;
    // t2/t3 overlap inside DFFT_PLAN_OVERLAP: the Y range of every destination is cut into `ycuts` sub-blocks; the last
    // X-plane part is exchanged sub-block by sub-block and the X pass of sub-block k runs while sub-block k+1 is in flight
    int                     ycuts = 1;
    std::vector<hipEvent_t> y_ev;
    // Padded work buffer for the Z <-> Y (and, on a single GPU, Y <-> X) intermediate of the fused pipeline: rows one
    // cache line longer than N2 and planes one more line apart, so that the column kernels' 128-byte segments -- N1 of
    // them one row pitch apart (Y pass), N0 of them one plane apart (X pass) -- do not all fall on the same memory
    // channels.  With power-of-two extents the natural strides (8 KiB, 4 MiB at 512^3 fp64) cost 11-15 % of the passes'
    // data rate (tools/membench4.hip, profiles/r02/README.md section 1).  Caller-visible buffers keep the reference layouts.
    void*                   wbuf = nullptr;
    SlabLayout              wl{0, 0};
    // an axis beyond the single-pass range (> 4096 points, dfft_long.hip): the plan runs the un-fused stage structure (only
    // contiguous rows and natural-layout columns need the four-step form then) and owns the scratch slab it needs
    bool                    long_axis = false;
    void*                   lbuf = nullptr;
    // Placement of the hand-over buffer (dfft_plan_tune).  The X pass runs 5-8 % faster when the buffer it reads and the
    // buffer it writes lie in different regions of the device's physical memory (regions are 4 ... 70 GiB long, consecutive
    // allocations usually share one; profiles/r03/README.md section 1, tools/xprobe.hip), so tuning times the X-pass kernel
    // alone on candidate allocations made one after the other -- all kept alive, so that each moves the next one on -- until
    // one behaves differently, and keeps the one on which it ran fastest.
    // t0 as one persistent launch (dfft_zy.hip) instead of two launches per cache chunk: single-GPU fused plans in fp64 whose Y and
    // Z lengths the kernel is built for.  zy_ctl: its control block (ticket counter, per-plane counters, error word).
    ZyCtl*                  zy_ctl = nullptr;
    // the plan's shape and flags select the one-launch stage (whether or not THIS device could allocate its control block): the same
    // on every device of a communicator, so it -- and not zy_ctl -- decides who takes part in the agreement round of dfft_execute
    bool                    zy_eligible = false;
    unsigned*               zy_err = nullptr;   // pinned host word the kernel writes ZY_ERR_* to when it gives up (read without a copy)
    unsigned                zy_spin_polls = 0;  // bound of a consumer unit's wait (polls; DFFT_ZY_SPIN_POLLS)
    int                     zy_fault = 0;       // DFFT_ZY_FAULT=n (test hook): launch number n of the stage waits for producers that never come
    unsigned                zy_launches = 0;
    bool                    zy_on = false;
    unsigned                zy_ticket = 0;  // value of the control block's ticket counter when the next launch starts
    unsigned                zy_execs = 0, zy_cur = 0;  // executes that have used the stage; index of the current one (per-plane counters)
    // overlapped forward plans: ONE launch of the stage over the whole slab that counts, per X-plane part, the column units whose results
    // are in memory (dfft_zy.hip, SIG); the exchange stream waits for a part's count and ships it while the launch computes the next part
    unsigned*               zy_part_done = nullptr;    // device: one counter per part (never reset: targets run on from execute to execute)
    unsigned                zy_sig_execs = 0;
    bool                    zy_lazy = false;           // lazy-publish form of the one-launch kernel (un-packed launches; DFFT_ZY_LAZY=0: eager)
    bool                    zy_inv_rows_first = true;  // backward single-GPU plans: inverse stage rows first (DFFT_ZY_INV_ROWS_FIRST=0: columns first)
    int                     x_hints = 0;               // DFFT_X_VARIANT when the plan was created: FFT_HINT_HALF_PREFETCH / _EARLY_WAIT
    int                     grid_x = 0, grid_y = 0, grid_z = 0;  // DFFT_X_GRID / DFFT_Y_GRID / DFFT_Z_GRID when the plan was created (0 = no cap)
    // Rows of the exchange buffers rotated by rot_elems elements per X plane (RotMap, dfft_kernels.h): P > 1 fused plans whose
    // received planes are a power-of-two distance apart.  0 = off.
    int                     rot_elems = 0;
    std::vector<float>      w_ms;            // report: X-pass time of every candidate tried (w_ms[w_kept] is the kept one)
    int                     w_kept = -1;
    float                   w_final_ms = 0.f;  // the kept candidate re-timed after the others were freed
};

static int fill_exchange(dfft_plan_s* p, ExchangeDesc& x, int direction) {
    const int       P = p->P, me = p->me;
    const long long n2 = p->N[2];
    x.dtype = p->dtype;
    x.direction = direction;
    x.P = P;
    x.me = me;
    x.scount.assign(P, 0);
    x.soffset.assign(P, 0);
    x.rcount.assign(P, 0);
    x.roffset.assign(P, 0);
    x.doffset.assign(P, 0);
    x.xsize.assign(P, 0);
    x.ysize.assign(P, 0);
    x.n2 = n2;
    for (int q = 0; q < P; ++q) {
        x.xsize[q] = p->sx.size(q);
        x.ysize[q] = p->sy.size(q);
    }
    for (int q = 0; q < P; ++q) {
        if (direction == DFFT_FORWARD) {
            // chunk(me -> q) = x in me's slab, y in q's slab          
            x.scount[q] = p->sx.size(me) * p->sy.size(q) * n2;
            x.soffset[q] = (long long)q * p->sx.size(me) * p->sy.blk * n2;   // packed [d][xl][yl_d][N2]
            x.rcount[q] = p->sx.size(q) * p->sy.size(me) * n2;
            x.roffset[q] = p->sx.start(q) * p->sy.size(me) * n2;              // [x][yl_me][N2], x = q*xl + xi
            x.doffset[q] = p->sx.start(me) * p->sy.size(q) * n2;
        } else {
            // chunk(me -> q) = x in q's slab, y in me's slab
            x.scount[q] = p->sx.size(q) * p->sy.size(me) * n2;
            x.soffset[q] = p->sx.start(q) * p->sy.size(me) * n2;
            x.rcount[q] = p->sx.size(me) * p->sy.size(q) * n2;
            x.roffset[q] = (long long)q * p->sx.size(me) * p->sy.blk * n2;
            x.doffset[q] = (long long)me * p->sx.size(q) * p->sy.blk * n2;
        }
    }
    return DFFT_OK;
}

// The packed side of the Y pass: [d][xs][yl_d][N2]; with Y sub-blocks (ycuts > 1, even X and Y splits only) [k][d][xs][yl/ycuts][N2]:
// sub-block k of the send buffer is exactly the region the X pass of sub-block k overwrites with its result, so a result never
// lands on send data that is still in flight (execute_forward).  tile->a_stride = distance between consecutive X planes in a block.
static void packed_map(const dfft_plan_s* p, AxisMap* packed, TileMap* tile) {
    const long long n2 = p->N[2];
    const long long ysub = p->sy.blk / p->ycuts;
    packed->blk = (int)ysub;
    packed->nblk = p->P * p->ycuts;
    packed->blk_stride = p->xs * ysub * n2;
    packed->sub = p->ycuts;
    packed->sub_stride = (long long)p->P * p->xs * ysub * n2;
    packed->stride = n2;
    packed->cstride = 1;
    packed->last_delta = p->ycuts > 1 ? 0 : (p->sy.size(p->P - 1) - p->sy.blk) * n2;
    *tile = TileMap{ysub * n2, 1};
}

// Y pass.  Natural side: [xs][N1][N2].  Packed side: [d][xs][yl_d][N2].
// lay_in / lay_out: layout of the natural side(s) (nullptr = {N2, N1*N2}); ignored for a packed side.
static int launch_y(dfft_plan_s* p, const void* in, void* out, bool packed_side_is_out, bool use_packed, long long x0,
                    long long nx, int hints = 0, const SlabLayout* lay_in = nullptr, const SlabLayout* lay_out = nullptr) {
    const int       n1 = (int)p->N[1];
    const long long n2 = p->N[2];
    if (n1 > 4096) {  // four-step form: natural layout on both sides only (long-axis plans run un-fused)
        auto natural = [&](const SlabLayout* l) { return !l || (l->pitch == n2 && l->plane == (long long)n1 * n2); };
        if (use_packed || !natural(lay_in) || !natural(lay_out)) return fail(DFFT_EINVAL, "Y pass: a long axis needs the natural layout");
        const size_t off = (size_t)x0 * n1 * n2 * elem_bytes(p->dtype);
        return long_fft((const char*)in + off, (char*)out + off, n1, n2, nx, p->dtype, p->direction, 1.0, p->lbuf, p->stream);
    }
    const void*     tw = nullptr;
    int             rc = get_twiddles(n1, p->dtype, &tw);
    if (rc) return rc;
    FftLaunch L;
    std::memset(&L, 0, sizeof(L));
    L.dtype = p->dtype;
    L.n = n1;
    L.dir = p->direction;
    L.cols = 1;
    L.in = in;
    L.out = out;
    L.tw = tw;
    const SlabLayout nat{n2, (long long)n1 * n2};
    const SlabLayout li = lay_in ? *lay_in : nat, lo = lay_out ? *lay_out : nat;
    AxisMap packed;
    TileMap pk_tile;
    packed_map(p, &packed, &pk_tile);
    L.imap = plain_axis(n1, li.pitch, 1);
    L.itile = TileMap{li.plane, 1};
    L.omap = plain_axis(n1, lo.pitch, 1);
    L.otile = TileMap{lo.plane, 1};
    if (use_packed) {
        if (packed_side_is_out) {
            L.omap = packed;
            L.otile = pk_tile;
        } else {
            L.imap = packed;
            L.itile = pk_tile;
        }
    }
    L.na = nx;
    L.a_first = x0;
    L.hints = hints;
    L.ncols = (int)n2;
    if (use_packed && p->rot_elems > 0) {  // the packed side's rows are rotated by the plane's global index
        L.rot.rot = p->rot_elems;
        L.rot.mask = (int)n2 - 1;
        L.rot.a0 = (int)p->sx.start(p->me);
        (packed_side_is_out ? L.rot.out_mode : L.rot.in_mode) = 1;
    }
    L.grid_limit = p->grid_y;
    return check_launch(launch_fft(L, p->stream), "Y pass");
}

// X pass.  Slab side: [N0][ys][N2] (x slowest).  Transposed side: [ys][N2][N0] (kx fastest).
// keep_slab: store [x][ys][N2] again instead of the transposed [ys][N2][kx] (natural-order plans)
// ys_part > 0: only a [N0][ys_part][N2] sub-slab (in/out already point at it)
// slab_lay: layout of the slab side when it is the plan's padded work buffer (single GPU: ys == N1), else [x][ys][N2]
static int launch_x(dfft_plan_s* p, const void* in, void* out, bool keep_slab = false, long long ys_part = 0,
                    const SlabLayout* slab_lay = nullptr) {
    const int       n0 = (int)p->N[0];
    const long long n2 = p->N[2];
    const long long ys = ys_part > 0 ? ys_part : p->ys;
    const void*     tw = nullptr;
    int             rc = get_twiddles(n0, p->dtype, &tw);
    if (rc) return rc;
    FftLaunch L;
    std::memset(&L, 0, sizeof(L));
    L.dtype = p->dtype;
    L.n = n0;
    L.dir = p->direction;
    L.cols = 1;
    L.in = in;
    L.out = out;
    L.tw = tw;
    AxisMap slab = plain_axis(n0, slab_lay ? slab_lay->plane : ys * n2, 1);
    TileMap slab_tile{slab_lay ? slab_lay->pitch : n2, 1};
    AxisMap tr = plain_axis(n0, 1, n0);
    TileMap tr_tile{n2 * (long long)n0, (long long)n0};
    if (keep_slab) {
        L.imap = L.omap = slab;
        L.itile = L.otile = slab_tile;
    } else if (p->direction == DFFT_FORWARD) {
        L.imap = slab;
        L.itile = slab_tile;
        L.omap = tr;
        L.otile = tr_tile;
    } else {
        L.imap = tr;
        L.itile = tr_tile;
        L.omap = slab;
        L.otile = slab_tile;
    }
    L.na = ys;
    L.scale = p->scale;
    L.ncols = (int)n2;
    if (p->rot_elems > 0 && p->exch && !keep_slab && !slab_lay) {  // the slab side is an exchange buffer with rotated rows
        L.rot.rot = p->rot_elems;
        L.rot.mask = (int)n2 - 1;
        (p->direction == DFFT_FORWARD ? L.rot.in_mode : L.rot.out_mode) = 2;
    }
    L.grid_limit = p->grid_x;
    L.hints |= p->x_hints;
    return check_launch(launch_fft(L, p->stream), "X pass");
}

namespace {
struct StageClock {
    dfft_plan_s* p;
    bool         sync;
    int          idx = 0;
    std::chrono::steady_clock::time_point t;
    int begin() {
        if (sync) {
            DFFT_HIP_TRY(hipStreamSynchronize(p->stream));
            t = std::chrono::steady_clock::now();
        } else if (p->timed) {
            DFFT_HIP_TRY(hipEventRecord(p->ev[0], p->stream));
        }
        return DFFT_OK;
    }
    int end_stage() {
        if (sync) {
            DFFT_HIP_TRY(hipStreamSynchronize(p->stream));
            auto now = std::chrono::steady_clock::now();
            p->host_t[idx] = std::chrono::duration<double>(now - t).count();
            t = now;
        } else if (p->timed) {
            DFFT_HIP_TRY(hipEventRecord(p->ev[idx + 1], p->stream));
        }
        ++idx;
        return DFFT_OK;
    }
};
}  // namespace

#define DFFT_TRY(stmt)        \
    do {                      \
        int rc_ = (stmt);     \
        if (rc_) return rc_;  \
    } while (0)

// Planes per phase of the one-launch YZ stage for a slab (or part) of nx planes: one that fits the 256 MiB Infinity Cache is ONE
// phase; larger ones are cut into equal phases of at most 230 MiB -- head-room instead of the largest possible size, see
// dfft_plan_create; a short last phase costs more than it saves.  DFFT_CHUNK_PLANES overrides.
// Forward PACKED launches (P > 1: the column units stream their results into the send buffer, so a phase is only READ from the cache) fill
// the cache instead -- 256 MiB: every phase boundary costs about 30 us of a rank's stage whatever the phase size (per rank at 512^3 fp64,
// P = 4, 128 planes: 2 phases of 64 planes 0.304 ms, 3 x 43 0.319, 4 x 32 0.394, 8 x 16 0.50, one phase of 512 MiB 0.377;
// profiles/r06/experiments/packed_phase_curve_512_P4.log), so fewer phases win as long as one still fits: 512^3 per rank at P = 2
// 5 x 52 -> 4 x 64 planes 0.597 -> 0.579 ms, at P = 4 3 x 43 -> 2 x 64 0.312 -> 0.300; config 4 at P = 2 14 -> 13 phases 2.03 -> 1.98.
// (What a boundary costs is the window in which row and column units run side by side: an interleaved ticket order -- rows of phase s
// alternating plane by plane with the columns of phase s - 1, no boundaries at all -- was built and measured at 0.335-0.46 ms against 0.309
// there, 1.42-1.82 against 1.155 on one GPU; profiles/r06/experiments/zy_interleaved_order.log.  Homogeneous phases, as few as fit.)
static long long zy_phase_planes(const dfft_plan_s* p, long long nx, bool packed_fwd = false) {
    static const long long forced = [] {
        const char* e = getenv("DFFT_CHUNK_PLANES");
        return e ? atoll(e) : 0ll;
    }();
    const long long plane_b = p->N[1] * p->N[2] * (long long)elem_bytes(p->dtype);
    if (forced > 0) return std::min(forced, nx);
    if (nx * plane_b <= (256ll << 20)) return nx;
    const long long fit = std::max(1ll, ((packed_fwd ? 256ll : 230ll) << 20) / plane_b), nch = (nx + fit - 1) / fit;
    return (nx + nch - 1) / nch;
}

// t0 (or its inverse) of planes [x0, x0 + nx) as one launch (dfft_zy.hip).
//   forward : Z rows src -> w, Y columns w -> w in place, or (packed) w -> the packed send layout in `other`
//   backward: Y columns w -> w in place, or (packed) the packed receive layout in `other` -> w; then Z rows w -> dst
// structure: 0 = the plan's direction; +1 on a backward plan = the inverse stage rows first (src = the hand-over buffer, w = the result buffer)
static int launch_zy_stage(dfft_plan_s* p, const void* src, void* w, long long w_plane, void* dst, void* other, bool packed, long long x0,
                           long long nx, int structure = 0, bool sig = false) {
    if (nx <= 0) return DFFT_OK;
    const void *twz = nullptr, *twy = nullptr;
    DFFT_TRY(get_twiddles((int)p->N[2], p->dtype, &twz));
    DFFT_TRY(get_twiddles((int)p->N[1], p->dtype, &twy));
    ZyLaunch L;
    std::memset(&L, 0, sizeof(L));
    L.dtype = p->dtype;
    L.n1 = (int)p->N[1];
    L.n2 = (int)p->N[2];
    L.dir = structure ? structure : p->direction;
    L.sign = p->direction;
    L.src = src;
    L.w = w;
    L.dst = dst;
    L.src_plane = L.dst_plane = p->N[1] * p->N[2];
    if (structure > 0 && p->direction < 0) L.src_plane = p->wl.plane;  // rows read the (plane-padded) hand-over buffer
    L.w_plane = w_plane; DFFT_ZY_SET_PITCH(L, (p->wbuf && w == p->wbuf) ? p->wl.pitch : p->N[2]);
    L.plane0 = x0;
    L.nplanes = nx;
    L.chunk = zy_phase_planes(p, nx, packed && p->direction == DFFT_FORWARD);
    L.ctl = p->zy_ctl;
    L.twz = twz;
    L.twy = twy;
    L.lazy = p->zy_lazy ? 1 : 0;
    L.spin_polls = p->zy_spin_polls;
    if (hipHostGetDevicePointer((void**)&L.err_host, p->zy_err, 0) != hipSuccess) return fail(DFFT_EHIP, "one-launch YZ stage: no device pointer for the error word");
    if (packed) {
        TileMap tile;
        packed_map(p, &L.pk, &tile);
        L.packed = 1;
        L.pk_plane = tile.a_stride;
        if (p->direction == DFFT_FORWARD) L.dst = other;
        else L.src = other;
        if (p->rot_elems > 0) {
            L.rot.rot = p->rot_elems;
            L.rot.mask = (int)p->N[2] - 1;
            L.rot.a0 = (int)p->sx.start(p->me);
        }
    }
    // the control block's counters run on from launch to launch (no reset between transforms): the ticket counter by what every
    // launch consumes, the per-plane counters by the producers of one plane per execute
    unsigned producers = 0;
    (void)zy_units_per_plane(L.n1, L.n2, L.dir, L.packed, &producers);
    if (x0 == 0) p->zy_cur = p->zy_execs++;
    L.ticket_base = p->zy_ticket;
    L.done_base = p->zy_cur * producers;
    if (sig) {  // whole-slab launch of the overlapped pipeline: count finished column units per X-plane part
        L.part_done = p->zy_part_done;
        L.part_planes = p->part_planes;
    }
    if (p->zy_fault > 0 && ++p->zy_launches == (unsigned)p->zy_fault) L.fault = 1;  // test hook: this launch's consumers can never start
    p->zy_ticket += zy_tickets(L.n1, L.n2, L.dir, L.packed, L.nplanes, L.chunk);
    return check_launch(launch_zy(L, p->stream), "one-launch YZ stage");
}

// Did a launch of the one-launch stage give up (a consumer unit exhausted its polls, or a launch found the control block's counters
// out of step)?  The kernel writes the reason to a pinned host word, so this costs one host load and can run anywhere -- before
// an execute is queued, after a drained one, in sync / stage_times / destroy.  The word is sticky on the device side (every later
// launch on the block returns at once), so nothing computed after the failure can pass for a result; the stage is switched off for
// good here and the plan continues on two launches per chunk.
static int zy_check(dfft_plan_s* p) {
    if (!p->zy_on || !p->zy_err) return DFFT_OK;
    const unsigned err = *(volatile unsigned*)p->zy_err;
    if (err == 0) return DFFT_OK;
    p->zy_on = false;  // two launches per chunk from now on
    return fail(DFFT_EHIP, std::string("the one-launch YZ stage gave up (") +
                               (err == ZY_ERR_TIMEOUT ? "a column / row unit exhausted its polls waiting for its plane's producers"
                                                      : "its ticket counter was out of step with the host's") +
                               "); the results of that execute and of every execute queued behind it are invalid, later executes of this plan "
                               "use the two-launch stage (DFFT_T0_ONE_LAUNCH=0 selects it from the start)");
}

static int execute_forward(dfft_plan_s* p, bool sync) {
    const bool      fused = !(p->flags & DFFT_PLAN_UNFUSED);
    const long long n1 = p->N[1], n2 = p->N[2], n0 = p->N[0];
    StageClock      clk{p, sync};
    DFFT_TRY(clk.begin());
    // ---- t0: 2D YZ FFT of every owned plane ----
    // The Z and Y passes run chunk by chunk over groups of planes that fit the 256 MiB Infinity Cache, so the Y pass
    // reads what the Z pass just wrote from cache instead of HBM (measured 1.73 -> 1.31..1.44 ms at 512^3 fp64).
    const void*     zsrc = (p->flags & DFFT_PLAN_INPUT_FROM_IN) ? p->in : p->buf1;
    const bool      y_packs = fused && p->exch;
    // where the Z pass puts its rows for the Y pass: the padded work buffer when the plan has one (fused pipelines)
    const SlabLayout nat{n2, n1 * n2};
    void*            zdst = (fused && p->wbuf) ? p->wbuf : p->buf1;
    const SlabLayout zl = (fused && p->wbuf) ? p->wl : nat;
    const SlabLayout *lz = (fused && p->wbuf) ? &zl : nullptr, *lnat = lz ? &nat : nullptr;  // row launches: layouts only when padded
    if (y_packs && (p->flags & DFFT_PLAN_OVERLAP) && p->part_planes > 0) {
        // ---- t0 pipelined against t2: the exchange of plane part k (stream2) runs while the Z+Y passes of part k+1
        // (stream) compute.  Any X-plane sub-range of the packed send layout is contiguous on both sides, so the parts
        // need no extra packing (dfft_exchange.cpp).  All ranks cut their slabs with the same part size.
        const bool rccl = comm_is_async(p->comm);
        const int  K = (int)((p->sx.blk + p->part_planes - 1) / p->part_planes);
        const int  YK = p->ycuts;
        // One launch of the YZ stage for ALL parts (round 6): the phases of the serial plan, no launch boundary and no 32-plane phase
        // per part; the stage counts, per part, the column units whose results are in memory, and the exchange stream waits for a
        // part's count (zy_part_wait_kernel) instead of an event behind a per-part launch.
        const bool one_for_all = p->zy_on && p->zy_part_done != nullptr;
        unsigned   sig_exec = 0, sig_units = 0;
        if (one_for_all) {
            unsigned producers = 0;
            sig_units = zy_units_per_plane((int)n1, (int)n2, +1, 1, &producers) - producers;  // column units per plane
            sig_exec = p->zy_sig_execs++;
            DFFT_TRY(launch_zy_stage(p, zsrc, zdst, zl.plane, nullptr, p->buf2, true, 0, p->xs, 0, true));
        }
        for (int k = 0; k < K; ++k) {
            long long x0, nx;
            part_range(p->xs, p->part_planes, k, &x0, &nx);
            if (one_for_all) {
                // (nothing to launch: the part is being computed by the launch above)
            } else if (nx > 0 && p->zy_on) {  // Z rows + packing Y columns of the part in one launch
                DFFT_TRY(launch_zy_stage(p, zsrc, zdst, zl.plane, nullptr, p->buf2, true, x0, nx));
            } else if (nx > 0) {
                DFFT_TRY(fft_rows(zsrc, zdst, (int)n2, nx * n1, p->dtype, p->direction, p->stream, x0 * n1,
                                  zsrc != zdst ? FFT_HINT_STREAM_IN : 0, 1.0, lnat, lz, lz ? n1 : 0));
                DFFT_TRY(launch_y(p, zdst, p->buf2, true, true, x0, nx, FFT_HINT_STREAM_OUT, &zl));
            }
            hipStream_t xs_ = rccl ? p->stream2 : p->stream;  // LOCAL: host-synchronising, same call sequence
            if (rccl && one_for_all) {
                // every execute adds nx * sig_units to the part's counter (all executes of a plan cut the same parts)
                unsigned* errw = nullptr;
                if (hipHostGetDevicePointer((void**)&errw, p->zy_err, 0) != hipSuccess) return fail(DFFT_EHIP, "one-launch YZ stage: no device pointer for the error word");
                // (no event between the streams: the wait kernel polls the counter, and the count can only be reached by THIS execute's
                // launch, which runs behind the previous execute's X pass on p->stream)
                if (nx > 0)
                    DFFT_TRY(check_launch(launch_zy_part_wait(p->zy_part_done + k, (sig_exec + 1u) * (unsigned)nx * sig_units, p->zy_ctl, errw, p->stream2),
                                          "part wait of the one-launch YZ stage"));
            } else if (rccl) {
                DFFT_HIP_TRY(hipEventRecord(p->part_ev[k], p->stream));
                DFFT_HIP_TRY(hipStreamWaitEvent(p->stream2, p->part_ev[k], 0));
            }
            if (k + 1 < K || YK == 1) {
                DFFT_TRY(comm_exchange_part(p->comm, p->xd, k, p->part_planes, xs_));
            } else {
                // last part: one exchange per Y sub-block, so the X pass can start on sub-block 0 while the others fly
                for (int y = 0; y < YK; ++y) {
                    DFFT_TRY(comm_exchange_part(p->comm, p->xd, k, p->part_planes, xs_, y));
                    if (rccl) DFFT_HIP_TRY(hipEventRecord(p->y_ev[y], p->stream2));
                }
            }
        }
        DFFT_TRY(clk.end_stage());
        DFFT_TRY(clk.end_stage());  // t1 folded into t0
        if (YK == 1) {
            if (rccl) {
                DFFT_HIP_TRY(hipEventRecord(p->join_ev, p->stream2));
                DFFT_HIP_TRY(hipStreamWaitEvent(p->stream, p->join_ev, 0));
            }
            DFFT_TRY(clk.end_stage());  // t2 = the part of the exchange that was not hidden behind t0
            DFFT_TRY(launch_x(p, p->rbuf, p->buf2));
        } else {
            // t3 pipelined against the tail of t2: sub-block y of the receive buffer is a complete [N0][ysub][N2] slab
            const long long ysub = p->ys / YK;
            const size_t    sub = (size_t)n0 * ysub * n2 * elem_bytes(p->dtype);
            for (int y = 0; y < YK; ++y) {
                if (rccl) DFFT_HIP_TRY(hipStreamWaitEvent(p->stream, p->y_ev[y], 0));
                if (y == 0) DFFT_TRY(clk.end_stage());  // t2 = exposed wait for the first sub-block
                DFFT_TRY(launch_x(p, (const char*)p->rbuf + y * sub, (char*)p->buf2 + y * sub, false, ysub));
            }
        }
        DFFT_TRY(clk.end_stage());
        return DFFT_OK;
    }
    const long long cp = p->chunk_planes > 0 ? p->chunk_planes : p->xs;
    // streaming hints whenever the stage works on cache-sized pieces of a slab too big to be cache-resident as a whole
    const bool      chunked = cp < p->xs || p->xs * n1 * n2 * (long long)elem_bytes(p->dtype) >= (64ll << 20);
    const bool      one_launch = p->zy_on && fused;
    if (one_launch) {  // the same chunk phases inside ONE persistent launch (dfft_zy.hip); P > 1: the Y side packs into the send buffer
        DFFT_TRY(launch_zy_stage(p, zsrc, zdst, zl.plane, nullptr, y_packs ? p->buf2 : nullptr, y_packs, 0, p->xs));
    }
    for (long long x0 = 0; !one_launch && x0 < p->xs; x0 += cp) {
        const long long nx = std::min(cp, p->xs - x0);
        DFFT_TRY(fft_rows(zsrc, zdst, (int)n2, nx * n1, p->dtype, p->direction, p->stream, x0 * n1,
                          (chunked && zsrc != zdst) ? FFT_HINT_STREAM_IN : 0, 1.0, lnat, lz, lz ? n1 : 0));
        if (!sync && p->timed && cp >= p->xs) DFFT_HIP_TRY(hipEventRecord(p->ev[5], p->stream));
        if (y_packs) DFFT_TRY(launch_y(p, zdst, p->buf2, true, true, x0, nx, chunked ? FFT_HINT_STREAM_OUT : 0, &zl));  // Y FFT + pack in one pass
        else DFFT_TRY(launch_y(p, zdst, zdst, true, false, x0, nx, 0, &zl, &zl));
    }
    if (y_packs) {
        DFFT_TRY(clk.end_stage());
        DFFT_TRY(clk.end_stage());  // t1 folded into t0
    } else {
        DFFT_TRY(clk.end_stage());
        // ---- t1: pack ----
        if (!fused) {
            hipError_t e = launch_pack(p->dtype, +1, p->buf1, p->buf2, (int)p->xs, (int)n1, (int)n2, (int)p->sy.blk,
                                       (int)p->sy.size(p->P - 1), p->P, p->stream);
            if (e != hipSuccess) return fail(DFFT_EHIP, std::string("pack: ") + hipGetErrorString(e));
        }
        DFFT_TRY(clk.end_stage());
    }
    // ---- t2: exchange ----
    const void* xsrc = p->exch ? p->buf1 : zdst;  // single GPU: the X pass reads what the Y pass left (possibly padded)
    if (p->exch) {
        DFFT_TRY(comm_exchange(p->comm, p->xd, p->stream));
    } else if (!fused) {
        // reference structure: full-size self copy bufferDev2 -> bufferDev1 (fft_mpi_3d_api.cpp:613-630)
        DFFT_HIP_TRY(hipMemcpyAsync(p->buf1, p->buf2, (size_t)p->xs * n1 * n2 * elem_bytes(p->dtype),
                                    hipMemcpyDeviceToDevice, p->stream));
    }
    DFFT_TRY(clk.end_stage());
    // ---- t3: X FFT (+ transpose to [yl][N2][N0]) ----
    if (fused) {
        DFFT_TRY(launch_x(p, xsrc, p->buf2, false, 0, (!p->exch && p->wbuf) ? &zl : nullptr));
    } else {
        hipError_t e = launch_transpose(p->dtype, p->buf1, p->buf2, n0, p->ys * n2, p->stream);
        if (e != hipSuccess) return fail(DFFT_EHIP, std::string("transpose: ") + hipGetErrorString(e));
        DFFT_TRY(fft_rows(p->buf2, p->buf2, (int)n0, p->ys * n2, p->dtype, p->direction, p->stream, 0, 0, p->scale));
    }
    DFFT_TRY(clk.end_stage());
    return DFFT_OK;
}

// DFFT_PLAN_NATURAL: input AND output are the caller's natural X-slab layout [xs][N1][N2] (the un-transposed output the
// reference declares but never implements, fft_mpi_local_size_3d / SURVEY 8f-2).  The same sequence serves both
// directions (the kernels take the sign): Z, Y(+pack) | exchange X->Y | X in slab layout | exchange Y->X | unpack.
static int execute_natural(dfft_plan_s* p, bool sync) {
    const long long n1 = p->N[1], n2 = p->N[2];
    StageClock      clk{p, sync};
    DFFT_TRY(clk.begin());
    const void*     zsrc = (p->flags & DFFT_PLAN_INPUT_FROM_IN) ? p->in : p->buf1;
    const long long cp = p->chunk_planes > 0 ? p->chunk_planes : p->xs;
    const bool      hinted = cp < p->xs || p->xs * n1 * n2 * (long long)elem_bytes(p->dtype) >= (64ll << 20);
    for (long long x0 = 0; x0 < p->xs; x0 += cp) {
        const long long nx = std::min(cp, p->xs - x0);
        DFFT_TRY(fft_rows(zsrc, p->buf1, (int)n2, nx * n1, p->dtype, p->direction, p->stream, x0 * n1,
                          (hinted && zsrc != p->buf1) ? FFT_HINT_STREAM_IN : 0));
        if (p->exch) DFFT_TRY(launch_y(p, p->buf1, p->buf2, true, true, x0, nx, hinted ? FFT_HINT_STREAM_OUT : 0));
        else DFFT_TRY(launch_y(p, p->buf1, p->buf1, true, false, x0, nx));
    }
    DFFT_TRY(clk.end_stage());
    DFFT_TRY(clk.end_stage());
    if (p->exch) DFFT_TRY(comm_exchange(p->comm, p->xd, p->stream));  // buf2 -> peers' buf1 = [x][ys][N2]
    DFFT_TRY(clk.end_stage());
    DFFT_TRY(launch_x(p, p->buf1, p->buf2, true));                      // [x][ys][N2] -> [kx][ys][N2]
    if (p->exch) {
        DFFT_TRY(comm_exchange(p->comm, p->xd2, p->stream));            // buf2 -> peers' rbuf = [src][xs][yl_src][N2]
        hipError_t e = launch_pack(p->dtype, -1, p->rbuf, p->buf2, (int)p->xs, (int)n1, (int)n2, (int)p->sy.blk,
                                   (int)p->sy.size(p->P - 1), p->P, p->stream);
        if (e != hipSuccess) return fail(DFFT_EHIP, std::string("unpack: ") + hipGetErrorString(e));
    }
    DFFT_TRY(clk.end_stage());
    return DFFT_OK;
}

static int execute_backward(dfft_plan_s* p, bool sync) {
    const bool      fused = !(p->flags & DFFT_PLAN_UNFUSED);
    const long long n1 = p->N[1], n2 = p->N[2], n0 = p->N[0];
    StageClock      clk{p, sync};
    DFFT_TRY(clk.begin());
    const void* src = (p->flags & DFFT_PLAN_INPUT_FROM_IN) ? p->in : p->buf1;
    // where the inverse Y pass leaves its columns for the inverse Z pass: the padded work buffer when the plan has one
    const SlabLayout nat{n2, n1 * n2};
    void*            ydst = (fused && p->wbuf) ? p->wbuf : p->buf2;
    const SlabLayout yl = (fused && p->wbuf) ? p->wl : nat;
    const SlabLayout *ly = (fused && p->wbuf) ? &yl : nullptr, *lnat = ly ? &nat : nullptr;
    if (fused && p->exch && (p->flags & DFFT_PLAN_OVERLAP) && p->part_planes > 0) {
        // ---- mirror image of the overlapped forward pipeline: the inverse X pass runs Y sub-block by sub-block into the
        // send buffer [k][x all][y in k][N2] and sub-block k is exchanged (stream2) while sub-block k+1 is transformed;
        // the last sub-block is exchanged X-plane part by part, and the Y+Z passes of part i start when part i has landed.
        const bool      rccl = comm_is_async(p->comm);
        const int       I = (int)((p->sx.blk + p->part_planes - 1) / p->part_planes);
        const int       YK = p->ycuts;
        const long long ysub = p->ys / YK;
        const size_t    sub = (size_t)n0 * ysub * n2 * elem_bytes(p->dtype);
        hipStream_t     xs_ = rccl ? p->stream2 : p->stream;  // LOCAL: host-synchronising, same call sequence
        for (int y = 0; y < YK; ++y) {
            if (YK == 1) DFFT_TRY(launch_x(p, src, p->rbuf));
            else DFFT_TRY(launch_x(p, (const char*)src + y * sub, (char*)p->rbuf + y * sub, false, ysub));
            if (rccl) {
                hipEvent_t done = YK == 1 ? p->join_ev : p->y_ev[y];
                DFFT_HIP_TRY(hipEventRecord(done, p->stream));
                DFFT_HIP_TRY(hipStreamWaitEvent(p->stream2, done, 0));
            }
            if (y + 1 < YK) {
                DFFT_TRY(comm_exchange_part(p->comm, p->xd, 0, p->sx.blk, xs_, y));  // all X planes of sub-block y
            } else {
                for (int i = 0; i < I; ++i) {
                    DFFT_TRY(comm_exchange_part(p->comm, p->xd, i, p->part_planes, xs_, YK == 1 ? -1 : y));
                    if (rccl) DFFT_HIP_TRY(hipEventRecord(p->part_ev[i], p->stream2));
                }
            }
        }
        DFFT_TRY(clk.end_stage());  // inverse X passes (the exchange of the earlier sub-blocks runs underneath)
        for (int i = 0; i < I; ++i) {
            if (rccl) DFFT_HIP_TRY(hipStreamWaitEvent(p->stream, p->part_ev[i], 0));
            if (i == 0) {
                DFFT_TRY(clk.end_stage());  // exposed part of the exchange
                DFFT_TRY(clk.end_stage());  // unpack folded into the Y pass
            }
            long long x0, nx;
            part_range(p->xs, p->part_planes, i, &x0, &nx);
            if (nx > 0 && p->zy_on) {  // unpacking Y columns + Z rows of the part in one launch
                DFFT_TRY(launch_zy_stage(p, nullptr, ydst, yl.plane, p->buf2, p->buf1, true, x0, nx));
            } else if (nx > 0) {
                DFFT_TRY(launch_y(p, p->buf1, ydst, false, true, x0, nx, FFT_HINT_STREAM_IN, nullptr, &yl));
                DFFT_TRY(fft_rows(ydst, p->buf2, (int)n2, nx * n1, p->dtype, p->direction, p->stream, x0 * n1, 0, 1.0, ly, lnat, ly ? n1 : 0));
            }
        }
        DFFT_TRY(clk.end_stage());
        return DFFT_OK;
    }
    // ---- inverse X FFT: [ys][N2][kx] -> [x][ys][N2] ----
    // (single GPU: straight into the padded work buffer -- the plane-strided 128-byte stores are the scattered side here)
    const bool xw = fused && !p->exch && p->wbuf;
    if (fused) {
        DFFT_TRY(launch_x(p, src, xw ? p->wbuf : p->buf2, false, 0, xw ? &yl : nullptr));
    } else {
        DFFT_TRY(fft_rows(src, p->buf1, (int)n0, p->ys * n2, p->dtype, p->direction, p->stream, 0, 0, p->scale));
        hipError_t e = launch_transpose(p->dtype, p->buf1, p->buf2, p->ys * n2, n0, p->stream);
        if (e != hipSuccess) return fail(DFFT_EHIP, std::string("transpose: ") + hipGetErrorString(e));
    }
    DFFT_TRY(clk.end_stage());
    // ---- exchange back ----
    void* ybuf = p->buf2;  // where the Y/Z passes run
    if (p->exch) {
        DFFT_TRY(comm_exchange(p->comm, p->xd, p->stream));  // buf2 -> peers' buf1 (packed [d][xs][yl_d][N2])
    } else if (!fused) {
        DFFT_HIP_TRY(hipMemcpyAsync(p->buf1, p->buf2, (size_t)p->xs * n1 * n2 * elem_bytes(p->dtype),
                                    hipMemcpyDeviceToDevice, p->stream));
    }
    DFFT_TRY(clk.end_stage());
    // ---- unpack + inverse Y, inverse Z ----
    if (fused) {
        DFFT_TRY(clk.end_stage());  // unpack folded into the Y pass
    } else {
        hipError_t e = launch_pack(p->dtype, -1, p->buf1, p->buf2, (int)p->xs, (int)n1, (int)n2, (int)p->sy.blk,
                                   (int)p->sy.size(p->P - 1), p->P, p->stream);
        if (e != hipSuccess) return fail(DFFT_EHIP, std::string("unpack: ") + hipGetErrorString(e));
        DFFT_TRY(clk.end_stage());
    }
    const bool      y_unpacks = fused && p->exch;
    const long long cp = p->chunk_planes > 0 ? p->chunk_planes : p->xs;
    // streaming hints whenever the stage works on cache-sized pieces of a slab too big to be cache-resident as a whole
    const bool      chunked = cp < p->xs || p->xs * n1 * n2 * (long long)elem_bytes(p->dtype) >= (64ll << 20);
    const bool one_launch = p->zy_on && fused;
    // Y columns (in place on the intermediate, or unpacking the receive buffer into it), then Z rows into the result: one launch
    // single-GPU plans with a hand-over buffer: rows first (dfft_zy.hip, SIGN) -- Z rows hand-over buffer -> result buffer, Y columns in place
    // on the result buffer; the transform is the same, its strided side moves from HBM reads to the cache-resident chunk
    // (the kernel's row-producing source path reads rows n2 apart: a hand-over buffer with padded ROWS -- the -DDFFT_ZY_ROW_PITCH=1 build
    // only -- keeps columns first)
    if (one_launch && p->zy_inv_rows_first && xw && p->zy_lazy && p->wl.pitch == n2) DFFT_TRY(launch_zy_stage(p, p->wbuf, ybuf, n1 * n2, nullptr, nullptr, false, 0, p->xs, +1));
    else if (one_launch) DFFT_TRY(launch_zy_stage(p, nullptr, fused ? ydst : ybuf, yl.plane, ybuf, y_unpacks ? p->buf1 : nullptr, y_unpacks, 0, p->xs));
    for (long long x0 = 0; !one_launch && x0 < p->xs; x0 += cp) {  // Y then Z per cache-sized chunk of planes (see execute_forward)
        const long long nx = std::min(cp, p->xs - x0);
        // single-GPU plans with a hand-over buffer, slab cut into cache chunks: Z rows first, like the one-launch stage above (round 6, last
        // session) -- rows hand-over buffer (HBM, contiguous) -> result buffer (natural layout, left in the Infinity Cache), then Y columns in
        // place on the result buffer: the strided side moves from HBM reads to the cache-resident chunk.  Inverse YZ stage 1024^3 fp32
        // 7.10 -> 5.93 ms, 2048 x 1024 x 512 fp64 13.08 -> 11.16, 512^3 fp32 0.89 -> 0.71; whole backward transform -7 ... -14 %
        // (profiles/r06/experiments/lib_ab_inverse_rows_first_chunks.log).  Same switches as the one-launch stage: DFFT_ZY_INV_ROWS_FIRST=0
        // or DFFT_ZY_LAZY=0 (the forms the bit-identity tests compare with each other) keep columns first.
        if (xw && cp < p->xs && p->zy_inv_rows_first && p->zy_lazy) {
            DFFT_TRY(fft_rows(p->wbuf, ybuf, (int)n2, nx * n1, p->dtype, p->direction, p->stream, x0 * n1, FFT_HINT_STREAM_IN, 1.0, ly, lnat, n1));
            DFFT_TRY(launch_y(p, ybuf, ybuf, false, false, x0, nx));
            continue;
        }
        if (y_unpacks) DFFT_TRY(launch_y(p, p->buf1, fused ? ydst : ybuf, false, true, x0, nx, chunked ? FFT_HINT_STREAM_IN : 0, nullptr, &yl));
        else if (xw) DFFT_TRY(launch_y(p, p->wbuf, p->wbuf, false, false, x0, nx, 0, &yl, &yl));
        else DFFT_TRY(launch_y(p, p->buf2, ybuf, false, false, x0, nx));
        if (!sync && p->timed && cp >= p->xs) DFFT_HIP_TRY(hipEventRecord(p->ev[5], p->stream));
        if (fused && p->wbuf) DFFT_TRY(fft_rows(ydst, ybuf, (int)n2, nx * n1, p->dtype, p->direction, p->stream, x0 * n1, 0, 1.0, ly, lnat, n1));
        else DFFT_TRY(fft_rows(ybuf, ybuf, (int)n2, nx * n1, p->dtype, p->direction, p->stream, x0 * n1));
    }
    DFFT_TRY(clk.end_stage());
    return DFFT_OK;
}

// Placement of the RECEIVE buffer of a P > 1 forward plan -- the P > 1 twin of dfft_plan_tune.  The X pass reads
// the receive buffer and writes the caller's `out`; like the hand-over buffer of a single-GPU plan the two normally come out of the
// same region of physical memory, the slow mode of any kernel that streams from one buffer into another.  The
// peers push into this buffer, so it has to be chosen BEFORE it is registered with the communicator: here, inside dfft_plan_create,
// for out-of-place fused forward plans whose received slab is larger than the Infinity Cache, and only while the buffer is still
// unknown to the peers (a pooled buffer of an IPC communicator was placed when its first plan was created).  Same walk as
// dfft_plan_tune: the plan's X-pass launches alone on the current buffer and on fresh allocations made one after the other, all
// kept until the end; stops at a confirmed 3.5 % gap; the probes write garbage into `out`, whose contents are set aside and put back.
// Every rank walks on its own (the registration that follows is the collective).
// Measured (round 5, per rank of 512^3 fp64 with the exchange switched off, profiles/r05/README.md section 6): X pass at P = 2 0.373 ->
// 0.341 ms (-8 %), at P = 4 0.176 vs 0.177 (nothing: no candidate in 5-8 differs from the first by more than noise once the pass runs
// inside the pipeline), config 4's rank at P = 8: 128 candidates within 3 % of each other.  A gain in one shape, plan-time cost and a
// transient footprint of many slabs in all of them, on a path that has never run on real links: OPT-IN, DFFT_TUNE_RECV=1 (at most
// DFFT_TUNE_TRIES, default 24, candidates).
// Bytes of a buffer that is (or may be) registered as a receive buffer: the larger of the last / not-last device's element count, so
// that every rank of a communicator asks its pool for the same size (comm_recv_alloc: pooled entries are matched by size).
static size_t recv_buffer_bytes(const dfft_plan_s* p) {
    const long long a = dfft_max_count(p->N[0], p->N[1], p->N[2], p->P, 0), b = dfft_max_count(p->N[0], p->N[1], p->N[2], p->P, 1);
    return (size_t)std::max(a, b) * elem_bytes(p->dtype);
}
static int place_recv_buffer(dfft_plan_s* p) {
    const char* te = getenv("DFFT_TUNE_RECV");
    if (!(te && *te == '1')) return DFFT_OK;
    if (!p->exch || p->P < 2 || p->direction != DFFT_FORWARD || p->inplace || p->long_axis || (p->flags & (DFFT_PLAN_UNFUSED | DFFT_PLAN_NATURAL)))
        return DFFT_OK;
    const size_t eb = elem_bytes(p->dtype);
    if ((size_t)p->N[0] * p->ys * p->N[2] * eb <= ((size_t)256 << 20)) return DFFT_OK;  // a cache-resident slab has no placement
    const bool is_rbuf = p->xd.recvbuf == p->rbuf && p->rbuf;
    void*      cur = is_rbuf ? p->rbuf : p->buf1;
    if (!cur || !comm_recv_is_fresh(p->comm, cur)) return DFFT_OK;
    const size_t bytes = (size_t)p->max_count * eb;
    DFFT_HIP_TRY(hipDeviceSynchronize());
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
        (void)hipGetLastError();
        return DFFT_OK;
    }
    const bool  alone = total_b > 0 && free_b >= total_b / 10 * 9;
    int         max_tries = 24, pct = alone ? 70 : 25;
    const char* mt = getenv("DFFT_TUNE_TRIES");
    if (mt && atoi(mt) > 0) max_tries = atoi(mt);
    const char* pe = getenv("DFFT_TUNE_MEM_PCT");
    if (pe && atoi(pe) >= 1 && atoi(pe) <= 90) pct = atoi(pe);
    const size_t budget = free_b / 100 * (size_t)pct;
    void*        saved_out = nullptr;
    if (2 * bytes > budget || hipMalloc(&saved_out, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return DFFT_OK;
    }
    if (hipMemcpyAsync(saved_out, p->buf2, bytes, hipMemcpyDeviceToDevice, p->stream) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(saved_out);
        return DFFT_OK;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        (void)hipGetLastError();
        if (e0) (void)hipEventDestroy(e0);
        (void)hipStreamSynchronize(p->stream);
        (void)hipFree(saved_out);
        return DFFT_OK;
    }
    const long long ysub = p->ys / std::max(1, p->ycuts);
    const size_t    sub = (size_t)p->N[0] * ysub * p->N[2] * eb;
    auto x_launches = [&](void* cand) -> int {  // exactly what execute_forward queues for t3
        if (p->part_planes > 0 && p->ycuts > 1) {
            for (int y = 0; y < p->ycuts; ++y) {
                const int rc = launch_x(p, (const char*)cand + y * sub, (char*)p->buf2 + y * sub, false, ysub);
                if (rc) return rc;
            }
            return DFFT_OK;
        }
        return launch_x(p, cand, p->buf2);
    };
    auto probe = [&](void* cand, float* ms_out) -> int {
        float t[5];
        for (int i = 0; i < 7; ++i) {
            if (hipEventRecord(e0, p->stream) != hipSuccess) return fail(DFFT_EHIP, "receive-buffer placement: event");
            const int rc = x_launches(cand);
            if (rc) return rc;
            float ms = 0.f;
            if (hipEventRecord(e1, p->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
                return fail(DFFT_EHIP, "receive-buffer placement: timing");
            if (i >= 2) t[i - 2] = ms;
        }
        std::sort(t, t + 5);
        *ms_out = t[2];
        return DFFT_OK;
    };
    std::vector<void*> cand(1, cur);
    p->w_ms.clear();
    float ms = 0.f;
    int   rc = probe(cur, &ms);
    if (rc == DFFT_OK) p->w_ms.push_back(ms);
    float  lo = ms, hi = ms;
    size_t used = bytes;
    while (rc == DFFT_OK && (int)cand.size() < max_tries && used + bytes <= budget) {
        if (lo < 0.97f * hi) {  // both behaviours seen?  confirm on a second timing of the extremes (dfft_plan_tune)
            int ilo = 0, ihi = 0;
            for (int i = 1; i < (int)p->w_ms.size(); ++i) {
                if (p->w_ms[i] < p->w_ms[ilo]) ilo = i;
                if (p->w_ms[i] > p->w_ms[ihi]) ihi = i;
            }
            for (int idx : {ilo, ihi}) {
                float again = 0.f;
                if (probe(cand[idx], &again) == DFFT_OK && again > 0.f) p->w_ms[idx] = std::min(p->w_ms[idx], again);
            }
            lo = *std::min_element(p->w_ms.begin(), p->w_ms.end());
            hi = *std::max_element(p->w_ms.begin(), p->w_ms.end());
            if (lo < 0.965f * hi) break;
        }
        void* nw = nullptr;
        if (hipMalloc(&nw, recv_buffer_bytes(p)) != hipSuccess) {  // (the size the pool was asked for: the candidate may take the buffer's place)
            (void)hipGetLastError();
            break;
        }
        used += bytes;
        cand.push_back(nw);
        rc = probe(nw, &ms);
        if (rc) break;
        p->w_ms.push_back(ms);
        lo = std::min(lo, ms);
        hi = std::max(hi, ms);
    }
    (void)hipStreamSynchronize(p->stream);
    int best = 0;
    for (int i = 1; i < (int)p->w_ms.size(); ++i)
        if (p->w_ms[i] < 0.985f * p->w_ms[best]) best = i;  // a later candidate must be clearly faster
    for (int i = 1; i < (int)cand.size(); ++i)
        if (i != best) (void)hipFree(cand[i]);
    if (best != 0) {
        // the chosen allocation takes the place of the plan's buffer; bufferDev1 carries the input captured at plan time
        if (!is_rbuf && hipMemcpyAsync(cand[best], p->in, bytes, hipMemcpyDeviceToDevice, p->stream) != hipSuccess) (void)hipGetLastError();
        (void)hipStreamSynchronize(p->stream);
        const int src = comm_recv_swap(p->comm, cur, cand[best]);
        if (src) rc = src;
        if (is_rbuf) p->rbuf = cand[best];
        else p->buf1 = cand[best];
        p->xd.recvbuf = cand[best];
    }
    p->w_kept = best;
    (void)hipMemcpyAsync(p->buf2, saved_out, bytes, hipMemcpyDeviceToDevice, p->stream);
    (void)hipStreamSynchronize(p->stream);
    (void)hipFree(saved_out);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (getenv("DFFT_DEBUG")) {
        fprintf(stderr, "[dfft] receive-buffer placement (rank %d): X pass", p->me);
        for (float v : p->w_ms) fprintf(stderr, " %.4f", v);
        fprintf(stderr, " ms, kept candidate %d\n", p->w_kept);
    }
    return rc;
}

extern "C" {

const char* dfft_version(void) { return "dfft-mi355x 0.1 (gfx950)"; }
const char* dfft_last_error(void) { return g_last_error.c_str(); }

int dfft_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int dfft_device_pci_bus_id(int device, char* buf, int len) {
    if (!buf || len < 16) return fail(DFFT_EINVAL, "dfft_device_pci_bus_id: buffer too small");
    if (device < 0) DFFT_HIP_TRY(hipGetDevice(&device));
    DFFT_HIP_TRY(hipDeviceGetPCIBusId(buf, len, device));
    return DFFT_OK;
}

int dfft_length_supported(long long n) {
    if (n <= 0 || n >= (1ll << 30)) return 0;
    if (fft_length_supported((int)n)) return 1;
    int a, b;
    return long_split(n, &a, &b) ? 1 : 0;  // two-pass (four-step) plans above 4096
}

int dfft_proper_device_count(const long long N[3], int ini_devices_in_rank, int nranks, int rank, int real_devices,
                             int* new_total, int* new_in_rank) {
    if (!N || !new_total || !new_in_rank || nranks < 1 || rank < 0 || rank >= nranks || ini_devices_in_rank < 1)
        return fail(DFFT_EINVAL, "dfft_proper_device_count: bad arguments");
    int ini = ini_devices_in_rank;
    if (real_devices >= 0 && ini > real_devices) ini = real_devices;  // fft_mpi_3d_api.cpp:236-239
    if (ini < 1) return fail(DFFT_ENOGPU, "dfft_proper_device_count: no device available");
    int total = ini * nranks, in_rank = ini;
    if (N[0] % total != 0) {  // :244-259
        const long long per = N[0] / total + 1;
        total = (int)(N[0] / per);
        if (N[0] % per != 0) total += 1;
        in_rank = total / nranks;
        const int rem = total % nranks;
        if (rem != 0 && rank < rem) in_rank += 1;
    }
    *new_total = total;
    *new_in_rank = in_rank;
    if (in_rank == 0) return fail(DFFT_EINVAL, "could not support this distribution of data");  // :266-269
    return DFFT_OK;
}

long long dfft_local_count(const long long N[3], int total_devices, int global_idx) {
    if (!N || total_devices < 1 || global_idx < 0 || global_idx >= total_devices) return -1;
    const Slab sx = make_slab(N[0], total_devices);
    return sx.size(global_idx) * N[1] * N[2];
}

long long dfft_max_count(long long n0, long long n1, long long n2, int total_devices, int is_last_device) {
    if (total_devices < 1) return -1;
    const Slab      sx = make_slab(n0, total_devices), sy = make_slab(n1, total_devices);
    const int       g = is_last_device ? total_devices - 1 : 0;
    const long long a = sx.size(g) * n1 * n2, b = n0 * sy.size(g) * n2;
    return a > b ? a : b;
}

int dfft_local_size(long long n0, long long n1, long long n2, int total_devices, int global_idx, long long* local_n0,
                    long long* local_0_start, long long* local_n1, long long* local_1_start) {
    (void)n2;
    if (total_devices < 1 || global_idx < 0 || global_idx >= total_devices)
        return fail(DFFT_EINVAL, "dfft_local_size: bad arguments");
    const Slab sx = make_slab(n0, total_devices), sy = make_slab(n1, total_devices);
    if (local_n0) *local_n0 = sx.size(global_idx);
    if (local_0_start) *local_0_start = sx.start(global_idx);
    if (local_n1) *local_n1 = sy.size(global_idx);
    if (local_1_start) *local_1_start = sy.start(global_idx);
    return DFFT_OK;
}

int dfft_exchange_layout(long long n0, long long n1, long long n2, int total_devices, int global_idx, int direction,
                         long long* scount, long long* soffset, long long* rcount, long long* roffset) {
    if (total_devices < 1 || global_idx < 0 || global_idx >= total_devices ||
        (direction != DFFT_FORWARD && direction != DFFT_BACKWARD))
        return fail(DFFT_EINVAL, "dfft_exchange_layout: bad arguments");
    dfft_plan_s tmp;
    tmp.N[0] = n0;
    tmp.N[1] = n1;
    tmp.N[2] = n2;
    tmp.P = total_devices;
    tmp.me = global_idx;
    tmp.direction = direction;
    tmp.dtype = DFFT_F64;
    tmp.sx = make_slab(n0, total_devices);
    tmp.sy = make_slab(n1, total_devices);
    if (tmp.sx.size(total_devices - 1) < 1 || tmp.sy.size(total_devices - 1) < 1)
        return fail(DFFT_EINVAL, "dfft_exchange_layout: last slab would be empty");
    fill_exchange(&tmp, tmp.xd, direction);
    for (int q = 0; q < total_devices; ++q) {
        if (scount) scount[q] = tmp.xd.scount[q];
        if (soffset) soffset[q] = tmp.xd.soffset[q];
        if (rcount) rcount[q] = tmp.xd.rcount[q];
        if (roffset) roffset[q] = tmp.xd.roffset[q];
    }
    return DFFT_OK;
}

int dfft_exchange_part_layout(long long n0, long long n1, long long n2, int total_devices, int global_idx, int direction,
                              long long part_planes, int part, int ycuts, int ycut, int max_msgs, int* peer,
                              long long* soffset, long long* scount, long long* roffset, long long* rcount) {
    if (total_devices < 1 || global_idx < 0 || global_idx >= total_devices || part_planes < 1 || part < 0 || ycuts < 1 ||
        ycut >= ycuts || max_msgs < 0 || (direction != DFFT_FORWARD && direction != DFFT_BACKWARD))
        return fail(DFFT_EINVAL, "dfft_exchange_part_layout: bad arguments");
    if (ycuts > 1 && (n0 % total_devices != 0 || n1 % total_devices != 0 || (n1 / total_devices) % ycuts != 0))
        return fail(DFFT_EINVAL, "dfft_exchange_part_layout: Y sub-blocks need even X and Y splits divisible by ycuts");
    dfft_plan_s tmp;
    tmp.N[0] = n0;
    tmp.N[1] = n1;
    tmp.N[2] = n2;
    tmp.P = total_devices;
    tmp.me = global_idx;
    tmp.direction = direction;
    tmp.dtype = DFFT_F64;
    tmp.sx = make_slab(n0, total_devices);
    tmp.sy = make_slab(n1, total_devices);
    if (tmp.sx.size(total_devices - 1) < 1 || tmp.sy.size(total_devices - 1) < 1)
        return fail(DFFT_EINVAL, "dfft_exchange_part_layout: last slab would be empty");
    fill_exchange(&tmp, tmp.xd, direction);
    tmp.xd.ycuts = ycuts;
    std::vector<int>       pe;
    std::vector<long long> so, sc, ro, rc;
    comm_part_messages(tmp.xd, part, part_planes, ycut, pe, so, sc, ro, rc);
    const int n = (int)pe.size();
    if (n > max_msgs) return fail(DFFT_EINVAL, "dfft_exchange_part_layout: more messages than max_msgs");
    for (int i = 0; i < n; ++i) {
        if (peer) peer[i] = pe[i];
        if (soffset) soffset[i] = so[i];
        if (scount) scount[i] = sc[i];
        if (roffset) roffset[i] = ro[i];
        if (rcount) rcount[i] = rc[i];
    }
    return n;
}

void* dfft_alloc(long long count, int dtype, int flag) {
    if (count < 0 || (dtype != DFFT_F64 && dtype != DFFT_F32)) {
        set_error("dfft_alloc: bad arguments");
        return nullptr;
    }
    const size_t bytes = (size_t)count * elem_bytes(dtype);
    void*        p = nullptr;
    if (flag == DFFT_ALLOC_HOST) {
        p = malloc(bytes ? bytes : 1);
    } else if (flag == DFFT_ALLOC_DEV) {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
        if (e != hipSuccess) {
            set_error(std::string("hipMalloc failed: ") + hipGetErrorString(e));
            return nullptr;
        }
    } else {
        set_error("Fail to allocate memory!");  // fft_mpi_3d_api.cpp:226
    }
    return p;
}

int dfft_free(void* p, int flag) {
    if (!p) return DFFT_OK;
    if (flag == DFFT_ALLOC_HOST) {
        free(p);
        return DFFT_OK;
    }
    DFFT_HIP_TRY(hipFree(p));
    return DFFT_OK;
}

int dfft_plan_create(dfft_plan_t* plan, long long n0, long long n1, long long n2, int dtype, int direction, void* in,
                     void* out, dfft_comm_t comm, int global_idx, int total_devices, unsigned flags) {
    if (!plan || !in) return fail(DFFT_EINVAL, "dfft_plan_create: null plan/in");
    if (n0 < 1 || n1 < 1 || n2 < 1) return fail(DFFT_EINVAL, "dfft_plan_create: sizes must be positive");
    if (dtype != DFFT_F64 && dtype != DFFT_F32) return fail(DFFT_EINVAL, "dfft_plan_create: dtype");
    if (direction != DFFT_FORWARD && direction != DFFT_BACKWARD) return fail(DFFT_EINVAL, "dfft_plan_create: direction");
    if (total_devices < 1 || global_idx < 0 || global_idx >= total_devices)
        return fail(DFFT_EINVAL, "dfft_plan_create: device index");
    if (total_devices > 1 && !comm) return fail(DFFT_EINVAL, "dfft_plan_create: a communicator is required for P > 1");
    if (comm && comm_size(comm) != total_devices) return fail(DFFT_EINVAL, "dfft_plan_create: communicator size != P");
    if (dfft_device_count() < 1) return fail(DFFT_ENOGPU, "dfft_plan_create: no HIP device visible (no CPU fallback)");
    for (long long n : {n0, n1, n2})
        if (!dfft_length_supported(n))
            return fail(DFFT_EUNSUPPORTED, "dfft_plan_create: FFT length " + std::to_string(n) + " has no gfx950 plan");

    const bool long_axis = n0 > 4096 || n1 > 4096 || n2 > 4096;
    if (long_axis && (flags & DFFT_PLAN_NATURAL))
        return fail(DFFT_EUNSUPPORTED, "dfft_plan_create: natural-order plans need single-pass axis lengths (<= 4096)");
    if (long_axis) flags = (flags | DFFT_PLAN_UNFUSED) & ~DFFT_PLAN_OVERLAP;  // the four-step axes run in the reference's stage structure
    trace("dfft_plan_create", n0 * 1000000 + n1 * 1000 + n2 % 1000, (long long)flags * 100 + total_devices);
    dfft_plan_s* p = new dfft_plan_s;
    p->long_axis = long_axis;
    p->N[0] = n0;
    p->N[1] = n1;
    p->N[2] = n2;
    p->dtype = dtype;
    p->direction = direction;
    p->P = total_devices;
    p->me = global_idx;
    p->flags = flags;
    p->is_last = (global_idx == total_devices - 1);
    {
        const char* fe = getenv("DFFT_FORCE_EXCHANGE");
        p->exch = total_devices > 1 || (comm && comm_kind(comm) == 1 && fe && *fe && *fe != '0');
    }
    p->sx = make_slab(n0, total_devices);
    p->sy = make_slab(n1, total_devices);
    p->xs = p->sx.size(global_idx);
    p->ys = p->sy.size(global_idx);
    p->comm = comm;
    p->grid_x = env_grid("DFFT_X_GRID");
    p->grid_y = env_grid("DFFT_Y_GRID");
    p->grid_z = env_grid("DFFT_Z_GRID");
    p->host_timed = false;
    for (double& t : p->host_t) t = 0;
    if (p->sx.size(total_devices - 1) < 1 || p->sy.size(total_devices - 1) < 1) {
        delete p;
        return fail(DFFT_EINVAL, "dfft_plan_create: slab decomposition leaves the last device empty");
    }
    p->max_count = dfft_max_count(n0, n1, n2, total_devices, p->is_last);
    if (p->max_count >= (1ll << 31)) {
        delete p;
        return fail(DFFT_EUNSUPPORTED, "dfft_plan_create: more than 2^31 elements per device");
    }
    p->in = in;
    p->out = out;
    p->inplace = (out == nullptr || out == in);  // fft_mpi_3d_api.cpp:68-75
    p->buf2 = p->inplace ? in : out;
    if (p->inplace && (flags & DFFT_PLAN_INPUT_FROM_IN)) {
        delete p;
        return fail(DFFT_EINVAL, "dfft_plan_create: DFFT_PLAN_INPUT_FROM_IN needs an out-of-place plan");
    }
    if ((flags & DFFT_PLAN_NATURAL) && (flags & DFFT_PLAN_UNFUSED)) {
        delete p;
        return fail(DFFT_EINVAL, "dfft_plan_create: DFFT_PLAN_NATURAL is a fused-pipeline option");
    }
    p->chunk_planes = 0;  // decided below, once the layout of the Z -> Y intermediate is known
    p->buf1 = nullptr;
    p->stream = nullptr;
    for (auto& e : p->ev) e = nullptr;
    const size_t bytes = (size_t)p->max_count * elem_bytes(dtype);
    hipError_t   e = hipGetDevice(&p->device);
    if (e == hipSuccess && comm && comm_kind(comm) == 0 && total_devices > 1) {
        // in-process multi-GPU (the reference's thread-per-GPU model): let this device push straight into its peers'
        // receive buffers over xGMI instead of staging peer copies through the host
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) == hipSuccess) {
            for (int d = 0; d < ndev; ++d) {
                int can = 0;
                if (d == p->device || hipDeviceCanAccessPeer(&can, p->device, d) != hipSuccess || !can) continue;
                hipError_t pe = hipDeviceEnablePeerAccess(d, 0);
                if (pe != hipSuccess) (void)hipGetLastError();  // already enabled / not supported: copies still work
            }
        }
    }
    // buffers that are (or may be) registered as receive buffers come from the communicator (pooled per key by IPC communicators)
    const std::string rkey = std::to_string(n0) + "x" + std::to_string(n1) + "x" + std::to_string(n2) + ":" + std::to_string(dtype) + ":" +
                             std::to_string(total_devices);
    const size_t rbytes = recv_buffer_bytes(p);  // rank-symmetric (>= bytes)
    if (e == hipSuccess && comm_recv_alloc(comm, rkey + ":b1", rbytes, &p->buf1) != DFFT_OK) e = hipErrorOutOfMemory;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
    // :77 input captured at plan time.  A device-to-device hipMemcpy on the null stream may return before the copy has run and is not
    // ordered with this plan's (non-blocking) stream -- and bufferDev1 is the RECEIVE buffer of the exchange: a copy that ran late
    // would overwrite data a peer had pushed already (round 5, tools/stall_hunt.py: the overlapped backward plan's first batch
    // differed from the serial result in 2 of 20 four-process runs).  So: everything the caller queued (whatever stream produced
    // `in`) is waited for, the copy runs on the plan's own stream, and plan creation returns when it has landed.
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyAsync(p->buf1, in, bytes, hipMemcpyDeviceToDevice, p->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
    for (auto& ev : p->ev)
        if (e == hipSuccess) e = hipEventCreate(&ev);
    // forward: always; backward: out-of-place plans that read the caller's `in` (the inverse X pass of sub-block k+1 must
    // not read a buffer the exchange of sub-block k is receiving into)
    if ((flags & DFFT_PLAN_OVERLAP) && p->exch && !(flags & (DFFT_PLAN_UNFUSED | DFFT_PLAN_NATURAL)) &&
        (direction == DFFT_FORWARD || (flags & DFFT_PLAN_INPUT_FROM_IN))) {
        // parts: DFFT_OVERLAP_PARTS (default 4) per slab, never larger than one Infinity-Cache chunk; derived from the
        // global block size ceil(N0/P) so that every rank cuts identically
        // How many: every part is a launch of its own (the one-launch YZ stage or a Z + Y pair) plus an exchange round, and short
        // launches pay their ramps -- four parts at every P taxed the local work of 512^3 fp64 by 30-40 % at P = 8, where a part was
        // 16 planes = 64 MiB = 40 us of work (t0 0.156 -> 0.207 ms per rank, profiles/r04/local_by_P.log), a third of what the overlap
        // can hide there.  So parts are sized, not counted: at least 128 MiB of slab each, between 2 and 4 of them (P = 8: 2 parts of
        // 128 MiB, P = 4 and below: 4), then capped by the cache chunk as before.  DFFT_OVERLAP_PARTS=n overrides.
        const long long plane_bytes = n1 * n2 * (long long)elem_bytes(dtype);
        long long       parts = std::min(4ll, std::max(2ll, (p->sx.blk * plane_bytes + (128ll << 20) - 1) / (128ll << 20)));
        const char*     pe = getenv("DFFT_OVERLAP_PARTS");
        if (pe && atoll(pe) > 0) parts = atoll(pe);
        long long pp = (p->sx.blk + parts - 1) / parts;
        const long long cache_planes = std::max(1ll, (256ll << 20) / plane_bytes);
        if (pp > cache_planes) pp = cache_planes;
        if (pp < 1) pp = 1;
        p->part_planes = pp;
        const int K = (int)((p->sx.blk + pp - 1) / pp);
        p->part_ev.assign(K, nullptr);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->stream2, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&p->join_ev, hipEventDisableTiming);
        for (auto& ev : p->part_ev)
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        // Y sub-blocks for the t2/t3 overlap (DFFT_OVERLAP_YPARTS, default 2; 1 = off): even Y split only
        long long   yk = 2;
        const char* ye = getenv("DFFT_OVERLAP_YPARTS");
        if (ye && atoll(ye) > 0) yk = atoll(ye);
        if (yk > 1 && n1 % total_devices == 0 && (n1 / total_devices) % yk == 0 && n0 % total_devices == 0) {
            p->ycuts = (int)yk;
            p->y_ev.assign(yk, nullptr);
            for (auto& ev : p->y_ev)
                if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        }
    }
    if (e != hipSuccess) {
        dfft_plan_destroy(p);
        return fail(DFFT_EHIP, std::string("dfft_plan_create: ") + hipGetErrorString(e));
    }
    const bool natural = (flags & DFFT_PLAN_NATURAL) != 0;
    fill_exchange(p, p->xd, natural ? DFFT_FORWARD : direction);
    p->xd.sendbuf = p->buf2;
    p->xd.recvbuf = p->buf1;
    p->xd.slot = p->xd2.slot = -1;  // registration ids, set below
    if (natural && p->exch) {
        // natural-order plans re-slab twice (X->Y for the X pass, Y->X to return to the caller's layout); the second
        // exchange receives the packed [src][xs][yl_src][N2] blocks into a buffer of its own and uses the other slot
        if (comm_recv_alloc(comm, rkey + ":nat", rbytes, &p->rbuf) != DFFT_OK) {
            dfft_plan_destroy(p);
            return fail(DFFT_EHIP, "dfft_plan_create: no memory for the second receive buffer");
        }
        fill_exchange(p, p->xd2, DFFT_BACKWARD);
        p->xd2.sendbuf = p->buf2;
        p->xd2.recvbuf = p->rbuf;
        int rc = comm_register(comm, global_idx, p->rbuf, p->device, &p->xd2.slot);
        if (rc) {
            dfft_plan_destroy(p);
            return rc;
        }
    }
    if (p->part_planes > 0) {
        // overlap mode: parts arrive while later planes are still being transformed in bufferDev1, so the exchange
        // needs a receive buffer of its own (one more slab in HBM; 288 GB makes that a non-issue)
        if (comm_recv_alloc(comm, rkey + ":rb", rbytes, &p->rbuf) != DFFT_OK) {
            dfft_plan_destroy(p);
            return fail(DFFT_EHIP, "dfft_plan_create: no memory for the receive buffer of the overlapped exchange");
        }
        if (direction == DFFT_FORWARD) p->xd.recvbuf = p->rbuf;  // Y pass -> out (send) -> rbuf (receive) -> X pass -> out
        else p->xd.sendbuf = p->rbuf;                            // in -> X pass -> rbuf (send) -> bufferDev1 -> Y,Z -> out
        p->xd.ycuts = p->ycuts;
    }
    // (the receive buffer is registered with the communicator further down, once its placement has been decided)
    // padded work buffer (dfft_plan_s::wbuf): fused, non-natural pipelines whose rows are whole cache lines and whose three
    // lengths run on the tuned kernels (the run-time-scheduled kernel keeps plain rows).  DFFT_PAD=0 switches it off.
    {
        const char* pe = getenv("DFFT_PAD");
        const long long line = 128 / (long long)elem_bytes(dtype);
        // Only where it pays (profiles/r02/experiments/padding_sweep.log): planes a multiple of 1 MiB apart -- other strides
        // spread over the channels by themselves (384^3: no gain) -- and slabs larger than the 256 MiB Infinity Cache
        // (cache-resident problems do not reach HBM's channels: 256^3 is 6 % slower with the extra buffer).
        const long long plane_b = n1 * n2 * (long long)elem_bytes(dtype);
        const bool      pays = (pe && *pe == '1') || (plane_b % (1ll << 20) == 0 && p->xs * plane_b > (256ll << 20));
        // (single GPU only: with an exchange the X pass reads the receive buffer, whose layout is the wire format)
        if (!(pe && *pe == '0') && pays && !p->exch && !(flags & (DFFT_PLAN_UNFUSED | DFFT_PLAN_NATURAL)) && (n2 % line) == 0 &&
            fft_length_tuned((int)n0) && fft_length_tuned((int)n1) && fft_length_tuned((int)n2)) {
            // lines of padding per row / per plane (tuning knobs DFFT_PAD_ROW, DFFT_PAD_PLANE; measured: profiles/r02)
            const char* pr = getenv("DFFT_PAD_ROW");
            const char* pp = getenv("DFFT_PAD_PLANE");
            const long long row_lines = pr ? atoll(pr) : 0, plane_lines = pp ? atoll(pp) : 3;
            p->wl.pitch = n2 + row_lines * line;
            p->wl.plane = n1 * p->wl.pitch + plane_lines * line;
            if (p->xs * p->wl.plane < (1ll << 31)) {
                e = slab_alloc(&p->wbuf, (size_t)p->xs * p->wl.plane * elem_bytes(dtype));
                if (e != hipSuccess) {  // an optimisation, not a requirement: run the natural layout in bufferDev1 instead
                    (void)hipGetLastError();
                    p->wbuf = nullptr;
                    if (getenv("DFFT_DEBUG")) fprintf(stderr, "[dfft] no memory for the padded hand-over buffer (%s): natural layout\n", hipGetErrorString(e));
                    e = hipSuccess;
                }
                if (getenv("DFFT_DEBUG")) fprintf(stderr, "[dfft] plan buffers: in %p out %p bufferDev1 %p work %p\n", in, out, p->buf1, p->wbuf);
            }
        }
    }
    {
        // Rotated rows in the exchange buffers (dfft_plan_s::rot_elems): the same remedy for the P > 1 pipeline that the padded
        // hand-over buffer is for the single-GPU one.  Every rank derives it from (N, P, precision, flags) alone -- both ends of
        // every message must agree.  Where: fused, even splits, tuned kernels on all three axes, rows of whole lines and a
        // power-of-two length, received planes a multiple of 128 KiB apart (other strides spread over the channels by
        // themselves).  DFFT_ROT=0 / 1 forces it off / on (wherever it is possible).
        const char*     re = getenv("DFFT_ROT");
        const long long S = (long long)elem_bytes(dtype);
        const bool      possible = p->exch && !(flags & (DFFT_PLAN_UNFUSED | DFFT_PLAN_NATURAL)) && !p->long_axis && n0 % total_devices == 0 &&
                              n1 % total_devices == 0 && (n2 & (n2 - 1)) == 0 && (n2 * S) % 128 == 0 && n2 * S >= 256 && n2 < (1ll << 30) &&
                              fft_length_tuned((int)n0) && fft_length_tuned((int)n1) && fft_length_tuned((int)n2);
        const long long ysub = p->ys / std::max(1, p->ycuts);
        // (round 6: 128 KiB, not 256 -- with received planes an odd multiple of 128 KiB apart the rotation is worth 7-10 % of the X pass in
        // five shapes of seven, nothing in one, -4 % in one: config 4's overlapped rank at P = 8, 384 KiB, 0.326 -> 0.299 ms;
        // profiles/r06/experiments/rot_pays_128k.log)
        const bool      pays = (ysub * n2 * S) % (128ll << 10) == 0;
        // How far: 3 cache lines per X plane (the pad of the single-GPU hand-over buffer, round 2) -- but 2 lines where the X axis is long
        // and the received planes lie at least 1 MiB apart.  Round 6 (profiles/r06/experiments/rot_lines_*.log, DFFT_ROT_LINES swept, four
        // plans per point): a tile of the X pass walks N0 segments (plane stride + rotation) apart; with 2048 planes 2 MiB apart (config 5's
        // rank at P = 8) 1 line 1.85 ms, 2 lines 1.67, 3 lines 1.89, 4 lines 1.98, 6 lines 1.67 -- odd multiples of 256 bytes spread the
        // walk over all channels, 384 bytes over two thirds of them, 512 over half.  2 against 3 lines: 1024^3 fp64 at P = 8 0.842 -> 0.737 ms,
        // 1024^3 fp32 at P = 4 0.801 -> 0.750, config 4 at P = 4 0.593 -> 0.569; no difference with 512 planes, and with planes
        // 512-768 KiB apart (512^3 overlapped at P = 4 / 8, config 4 at P = 8) 3 lines stay 2-10 % ahead.  DFFT_ROT_LINES=n overrides.
        if (possible && !(re && *re == '0') && (pays || (re && *re == '1'))) {
            int         lines = (n0 >= 1024 && ysub * n2 * S >= (1ll << 20)) ? 2 : 3;
            const char* rl = getenv("DFFT_ROT_LINES");
            if (rl && atoi(rl) > 0) lines = atoi(rl);
            p->rot_elems = (int)(lines * 128 / S);
        }
    }
    {
        // kernel-variant switches (A/B measurements; results are bit-identical either way), read when the plan is created:
        //   DFFT_X_VARIANT=half|full|fullearly  1024-point forward X pass: half-tile prefetch / whole-tile prefetch (default) / + early wait
        //   DFFT_X_VARIANT=early                512-point forward X pass: early wait for the prefetched tile
        //   DFFT_ZY_LAZY=0                      one-launch YZ stage (P = 1): the eager-publish kernel instead of the lazy one
        const char* xv = getenv("DFFT_X_VARIANT");
        if (xv && !strcmp(xv, "half")) p->x_hints = FFT_HINT_HALF_PREFETCH;
        else if (xv && (!strcmp(xv, "fullearly") || !strcmp(xv, "early"))) p->x_hints = FFT_HINT_EARLY_WAIT;
        const char* zl = getenv("DFFT_ZY_LAZY");
        p->zy_lazy = !(zl && *zl == '0');  // default on: t0 of 512^3 fp64 1.278 -> 1.163 ms (profiles/r03/experiments/variant_ab.log)
        const char* rf = getenv("DFFT_ZY_INV_ROWS_FIRST");
        p->zy_inv_rows_first = !(rf && *rf == '0');
    }
    if (comm) {
        // placement of the receive buffer (P > 1 twin of dfft_plan_tune), then its registration: after that the peers know the pointer
        int rc = place_recv_buffer(p);
        if (rc == DFFT_OK) rc = comm_register(comm, global_idx, p->xd.recvbuf, p->device, &p->xd.slot);  // nodeDataDev[loc] = bufferDev1, :80
        if (rc) {
            dfft_plan_destroy(p);
            return rc;
        }
    }
    {
        // one-launch t0 (dfft_zy.hip): where the kernel exists and the plan has the unpadded-row hand-over buffer it works on
        const char* oe = getenv("DFFT_T0_ONE_LAUNCH");
        const long long ysub = p->sy.blk / std::max(1, p->ycuts);
        const bool      single_ok = !p->exch && (!p->wbuf || p->wl.pitch == n2 || DFFT_ZY_ROW_PITCH);
        const bool      multi_ok = p->exch && n0 % total_devices == 0 && n1 % total_devices == 0 && n1 >= 8 && ysub % zy_col_threads((int)n1, p->zy_lazy ? 1 : 0) == 0;  // even splits; a
                                   // destination block is a whole number of the column unit's strides (the threads of one column FFT)
        // Where the stage is used by itself (DFFT_T0_ONE_LAUNCH=0: never; =1: wherever the kernel exists on single-GPU plans; =all:
        // P > 1 plans too).  Round 3 used it for 512 x 512-point planes on a single GPU only; since round 4
        //  * un-packed launches with a 256-point Y axis take column tiles of two cache lines (ZyTile, dfft_zy.hip: 512-thread
        //    workgroups and 64 KiB units again), and the stage wins on every plane shape it is built for -- t0 / back-to-back ms per
        //    transform, two launches per chunk -> one launch: 256^3 (BASELINE config 2) 0.169 / 0.2445 -> 0.154 / 0.2350, 512 x 256 x 256
        //    0.357 / 0.510 -> 0.305 / 0.463, 256 x 256 x 512 0.349 / 0.509 -> 0.307 / 0.469, 256 x 512 x 256 0.355 / 0.528 -> 0.309 / 0.470
        //    (profiles/r04/experiments/variant_ab_256_wide_tiles.log; with one-line tiles it lost: 0.170 vs 0.216 at 256^3);
        //  * the packed launches of P > 1 plans run the lazy-publish form too (it measured equal to two launches while it published
        //    eagerly): per rank at 512^3 fp64, exchange switched off, t0 / back-to-back P = 2 0.690 / 1.052 -> 0.630 / 0.94, P = 4
        //    0.343 / 0.512 -> 0.32 / 0.492, P = 8 0.174 / 0.255 -> 0.157 / 0.248 (experiments/lib_ab_lazy_packed.log) -- used by itself
        //    for 512 x 512-point planes (packed tiles stay one line wide, so planes with a 256-point axis keep two launches).
        //  * round 5: planes with a 768-point Y axis (BASELINE config 4's 768 x 512 planes; 12 points x 64 threads, twiddle table of the Y
        //    axis in LDS like the two-launch kernel's).  t0 / back-to-back ms, two launches per chunk -> one launch, 1024 x 768 x 512 fp64:
        //    P = 1 4.49 / 7.23 -> 3.99 / 6.73, per rank at P = 4 (192-row destination blocks) 1.12 / 1.84 -> 1.06 / 1.74 -- but at P = 8
        //    (96-row blocks, two offset tables per thread half) 0.57-0.60 / 0.91-0.93 -> 0.62 / 0.94 whatever the phase size
        //    (profiles/r05/experiments/variant_ab_768_final.log, c4_p8_one_launch_sweep.log): used by itself where a destination block is
        //    whole 64-row strides of the column unit, i.e. up to P = 4 for this axis.
        //  * round 6, overlapped forward plans: the ONE launch for all parts (zy_part_done below) also pays on 96-row destination blocks --
        //    config 4 per rank at P = 4 (two Y sub-blocks of 96 rows), overlapped t0 1.16 -> 1.05 ms against 1.03 for the serial plan
        //    (profiles/r06/experiments/c4_overlap_one_launch.log): what it replaces there is one launch pair PER PART, not per cache chunk.
        const bool      parts_fwd = p->part_planes > 0 && direction == DFFT_FORWARD && p->zy_lazy;
        const bool      multi_on = (n1 == 512 && n2 == 512) || (n1 == 768 && n2 == 512 && (ysub % 64 == 0 || parts_fwd)) || (oe && !strcmp(oe, "all"));
        if (!(oe && *oe == '0') && (single_ok || (multi_ok && multi_on)) && !(flags & (DFFT_PLAN_UNFUSED | DFFT_PLAN_NATURAL)) && !p->long_axis &&
            zy_supported(dtype, (int)n1, (int)n2) && p->xs <= ZY_MAX_PLANES) {
            p->zy_eligible = true;
            // zeroed ON THE PLAN'S STREAM and waited for: a memset on the null stream is asynchronous to the host and not ordered
            // with a non-blocking stream -- the first launch could start on uninitialised counters (seen once in the full test
            // suite, on recycled memory)
            if (hipMalloc((void**)&p->zy_ctl, sizeof(ZyCtl)) == hipSuccess && hipMemsetAsync(p->zy_ctl, 0, sizeof(ZyCtl), p->stream) == hipSuccess &&
                hipStreamSynchronize(p->stream) == hipSuccess && hipHostMalloc((void**)&p->zy_err, 128, hipHostMallocMapped) == hipSuccess) {
                *p->zy_err = 0u;
                p->zy_on = true;
                // a consumer's wait is bounded in polls (1-3 us each under load): seconds by default, so that time-slicing with
                // other processes, a profiler or a debugger cannot produce a spurious time-out, while a real dead-lock still ends
                const char* sp = getenv("DFFT_ZY_SPIN_POLLS");
                p->zy_spin_polls = sp && atoll(sp) > 0 ? (unsigned)std::min(atoll(sp), 0xffffffffll) : (4u << 20);
                const char* fe = getenv("DFFT_ZY_FAULT");
                p->zy_fault = fe ? atoi(fe) : 0;
                // overlapped forward plans: one launch for all parts (per-part counters of finished column units; the kernel addresses the
                // packed layout of a slab with 32-bit byte offsets there).  DFFT_ZY_PARTS_ONE_LAUNCH=0: one launch per part (round 5's form).
                const char* oa = getenv("DFFT_ZY_PARTS_ONE_LAUNCH");
                const int   K = p->part_planes > 0 ? (int)((p->sx.blk + p->part_planes - 1) / p->part_planes) : 0;
                if (p->exch && direction == DFFT_FORWARD && K > 0 && p->zy_lazy && !(oa && *oa == '0') && (size_t)p->max_count * elem_bytes(dtype) < ((size_t)1 << 32)) {
                    if (hipMalloc((void**)&p->zy_part_done, (size_t)K * sizeof(unsigned)) != hipSuccess ||
                        hipMemsetAsync(p->zy_part_done, 0, (size_t)K * sizeof(unsigned), p->stream) != hipSuccess || hipStreamSynchronize(p->stream) != hipSuccess) {
                        (void)hipGetLastError();
                        if (p->zy_part_done) (void)hipFree(p->zy_part_done);
                        p->zy_part_done = nullptr;
                    }
                }
            } else {
                (void)hipGetLastError();
            }
        }
    }
    {
        // Z+Y blocking for the 256 MiB Infinity Cache (MI355X_MICROARCH.md): the largest whole number of NATURAL planes that
        // fits, evened out over the chunks -- 8 x 64 planes at 512^3 fp64.  With the padded work buffer such a chunk is a few
        // KiB larger than the cache; measured (profiles/r02/experiments/chunk_planes_sweep.log, two boxes) that is still the
        // best size: t0 1.37-1.38 ms against 1.40-1.42 for 9 x 57 (sized on the padded planes), 60 or 63 planes, and the X pass
        // runs no slower behind it.  t0 follows the chunk count (~9 us per Z+Y launch pair) and is worst just above a
        // power-of-two size (40 planes: 1.63 ms); chunks of whole grid-stride rounds (56 / 60 planes) gain nothing.
        // Planes of 8 MiB and more (1024-point Y columns) keep one plane of head-room: 1024^3 15 instead of 16 planes t0 12.6 ->
        // 12.4 ms, 2048 x 1024 x 512 31 instead of 32 planes 12.5 -> 12.1 ms; smaller planes want the full count (512^3 fp64 64
        // planes, fp32 128 planes 0.80 -> 0.74 ms, 384^3 4 x 96; profiles/r02/experiments/chunk_shapes.log).
        // DFFT_CHUNK_MB=0 disables, =k overrides the capacity; DFFT_CHUNK_PLANES=n sets the chunk size directly (experiments).
        // The one-launch stage (dfft_zy.hip) has no launch boundaries to amortise and wants head-room instead: 9 phases of 57 planes
        // (228 MiB) run t0 of 512^3 fp64 in 1.213-1.216 ms, 8 x 64 (256 MiB + padding) in 1.297, 10 x 52 in 1.238, and any
        // count that leaves a short last phase loses (56 -> 9 x 56 + 8: 1.284; 62: 1.341; profiles/r03/experiments/
        // chunk_planes_one_launch.log) -- so: 230 MiB, evened out over the phases as before.
        long long   mb = p->zy_on ? 230 : 256;
        const char* ce = getenv("DFFT_CHUNK_MB");
        if (ce) mb = atoll(ce);
        const char* re = getenv("DFFT_CHUNK_RULE");  // 0: the rule of rounds 2-3 for every plan (A/B switch)
        const bool  y_streams_out = p->exch && !(flags & (DFFT_PLAN_UNFUSED | DFFT_PLAN_NATURAL));
        if (mb > 0 && !p->zy_on && y_streams_out && !(re && *re == '0')) {
            // Fused P > 1 plans on two launches per chunk (round 4, profiles/r04/experiments/chunk_rule.log): the Y pass packs its
            // results straight into the send buffer with streaming stores, so the chunk is only READ from the cache and may fill it
            // (no head-room), and the largest chunk whose column tiles are a whole number of grid rounds (one tile per CU and round)
            // beats evened-out chunks, a short last chunk included -- a launch's ramp is amortised over more bytes.  Per rank, t0 /
            // back-to-back ms: config 4 at P = 8 (128 planes of 6 MiB) 4 x 32 -> 3 x 40 + 8 planes 0.613 / 0.953 -> 0.585 / 0.924, at
            // P = 4 1.190 / 1.890 -> 1.150 / 1.848; config 5 at P = 8 (16 MiB planes) 15 -> 16 planes 3.298 / 5.157 -> 3.207 / 5.060;
            // 1024^3 fp64 at P = 8 1.602 / 2.563 -> 1.544 / 2.486.  Single-GPU plans, whose Y pass works in place on the chunk, lose
            // with it (512^3 on two launches 8 x 64 -> 9 x 60 planes t0 1.380 -> 1.405 ms, 512 x 2048 x 512 6.21 -> 6.80) and keep the
            // rule below.
            const long long eb = (long long)elem_bytes(dtype);
            long long       fit = std::max(1ll, (mb << 20) / (n1 * n2 * eb));
            const long long line = 128 / eb, cus = std::max(1ll, zy_grid());  // (CUs of the current device)
            if (n2 % line == 0) {
                const long long tiles_per_plane = n2 / line;
                const long long q = cus / std::gcd(cus, tiles_per_plane);  // planes per whole number of grid rounds
                if (q > 1 && fit >= 2 * q) fit -= fit % q;
            }
            p->chunk_planes = fit;
            // Overlapped plans on two launches per PART (no one-launch stage for the shape): a part is one Z + Y launch pair, so the same
            // rule sizes it -- when that does not change the number of parts (the overlap's granularity): config 4 per rank at P = 8,
            // 4 x 32 -> 3 x 40 + 8 planes (a 32-plane launch pair runs at the rate of a 32-plane chunk: serial t0 0.547 ms with 40-plane
            // chunks, 0.59 with 32, 0.616 for the four 32-plane parts; profiles/r06/experiments/c4_overlap_one_launch.log).  Decided from
            // rank-symmetric data only (the shape predicate zy_eligible, the global block size): every rank cuts the same parts.
            const long long blk = p->sx.blk, pp = p->part_planes;
            if (pp > 0 && !p->zy_eligible && !getenv("DFFT_OVERLAP_PARTS") && !ce && fit > pp && fit < blk && (blk + fit - 1) / fit == (blk + pp - 1) / pp)
                p->part_planes = fit;
        } else if (mb > 0) {
            const long long plane_b = n1 * n2 * (long long)elem_bytes(dtype);
            long long       fit = std::max(1ll, (mb << 20) / plane_b);
            if (plane_b >= (8ll << 20) && fit > 1) --fit;
            const long long nchunks = std::max(1ll, (p->xs + fit - 1) / fit);
            p->chunk_planes = (p->xs + nchunks - 1) / nchunks;
        }
        const char* cpe = getenv("DFFT_CHUNK_PLANES");
        if (cpe && atoll(cpe) > 0) p->chunk_planes = atoll(cpe);
        if (p->chunk_planes >= p->xs) p->chunk_planes = 0;
    }
    if (p->long_axis) {
        p->chunk_planes = 0;  // the four-step passes work on the whole slab
        e = hipMalloc(&p->lbuf, (size_t)p->max_count * elem_bytes(dtype));
        if (e != hipSuccess) {
            dfft_plan_destroy(p);
            return fail(DFFT_EHIP, std::string("dfft_plan_create: ") + hipGetErrorString(e));
        }
    }
    // warm the twiddle caches so execute never allocates
    for (long long n : {n0, n1, n2}) {
        if (n > 4096) continue;  // long axes: their factors' tables are built on first use
        const void* tw;
        int         rc = get_twiddles((int)n, dtype, &tw);
        if (rc) {
            dfft_plan_destroy(p);
            return rc;
        }
    }
    *plan = p;
    return DFFT_OK;
}

int dfft_plan_set_scale(dfft_plan_t plan, double s) {
    if (!plan || !(s == s) || s == 0.0) return fail(DFFT_EINVAL, "dfft_plan_set_scale: bad arguments");
    plan->scale = s;
    return DFFT_OK;
}

void* dfft_plan_buffer1(dfft_plan_t plan) { return plan ? plan->buf1 : nullptr; }
void* dfft_plan_result(dfft_plan_t plan) { return plan ? plan->buf2 : nullptr; }
void* dfft_plan_stream(dfft_plan_t plan) { return plan ? (void*)plan->stream : nullptr; }
void* dfft_plan_workbuf(dfft_plan_t plan, long long* bytes) {
    if (!plan || !plan->wbuf) return nullptr;
    if (bytes) *bytes = plan->xs * plan->wl.plane * (long long)elem_bytes(plan->dtype);
    return plan->wbuf;
}

int dfft_execute(dfft_plan_t plan, unsigned exec_flags) {
    if (!plan) return fail(DFFT_EINVAL, "dfft_execute: null plan");
    const bool sync = (exec_flags & DFFT_EXEC_SYNC_STAGES) != 0;
    plan->host_timed = sync;
    plan->timed = sync || !(exec_flags & DFFT_EXEC_NO_TIMING);
    if (!plan->timed && (exec_flags & DFFT_EXEC_PRINT)) return fail(DFFT_EINVAL, "dfft_execute: PRINT needs stage timing");
    // an earlier execute of the one-launch YZ stage that gave up and that nobody synchronised through the library (the pinned error
    // word is readable at any time): report it before anything else is queued; the stage is off from here on.  A plan with an
    // exchange queues this execute all the same (on the two-launch stage) and reports afterwards: returning early on ONE rank would
    // leave its peers waiting in the exchange for a partner that never comes.
    const int   zrc0 = zy_check(plan);
    std::string zmsg0 = zrc0 ? g_last_error : std::string();
    if (zrc0 && !plan->exch) return zrc0;
    auto run = [&]() {
        t_plan_scratch = plan->lbuf;
        t_plan_zgrid = plan->grid_z;
        const int r = (plan->flags & DFFT_PLAN_NATURAL) ? execute_natural(plan, sync)
                      : plan->direction == DFFT_FORWARD ? execute_forward(plan, sync)
                                                        : execute_backward(plan, sync);
        t_plan_scratch = nullptr;
        t_plan_zgrid = 0;
        return r;
    };
    const bool zy_used = plan->zy_on;
    if (plan->P > 1) trace("dfft_execute", plan->direction, exec_flags);
    int        rc = run();
    // host-synchronised executes have drained the stream (the reference-named wrapper always executes this way): a one-launch YZ
    // stage that gave up is seen here, before any timing is printed or any result used.  Where the pipeline has not written its own
    // input -- fused single-GPU plans that read `in` or bufferDev1 and work in the hand-over buffer -- the transform is simply run
    // again on two launches per chunk (zy_check has switched the stage off); otherwise the failure is the return code.
    // Who takes part in the agreement round is decided by rank-symmetric data only: the shape / flag predicate that selects
    // the stage (zy_eligible), not this device's allocation result or history -- and a device whose run() failed locally still enters
    // the round (contributing "failed"), so its peers are never left waiting in a collective one rank has skipped.
    const bool agree = sync && plan->exch && plan->zy_eligible && plan->direction == DFFT_FORWARD;
    if (rc && !agree) return rc;
    if (agree) {
        // (backward plans run the stage AFTER their exchange: a failure there spoils only the failing rank's own result, which that
        // rank reports itself below -- no collective needed, none paid)
        // P > 1: the failure is rank-local knowledge, but the exchange has already shipped this rank's invalid data to every peer.
        // Every device of the communicator learns of it here, returns an error and continues on the two-launch stage, so no rank
        // ever returns DFFT_OK for an execute that any rank knows to be invalid.
        const int   run_rc = rc;
        std::string run_msg = run_rc ? g_last_error : std::string();
        int         mine = (zrc0 || run_rc) ? 1 : 0;
        if (!mine && zy_used) {
            mine = zy_check(plan) ? 1 : 0;
            if (mine) zmsg0 = g_last_error;
        }
        int any = 0;
        rc = comm_agree_max(plan->comm, plan->me, mine, plan->stream, &any);
        if (run_rc) return fail(run_rc, run_msg);  // this device's own failure is its return code, whatever the round said
        if (rc) return rc;
        if (any) {
            plan->zy_on = false;
            return mine ? fail(DFFT_EHIP, zmsg0)
                        : fail(DFFT_EHIP, "the one-launch YZ stage gave up (or the execute failed) on another device of this communicator: the results of this "
                                          "execute are invalid on every device; later executes use the two-launch stage");
        }
    } else if (sync && zy_used) {
        rc = zy_check(plan);
        if (rc) {
            const void* src = (plan->flags & DFFT_PLAN_INPUT_FROM_IN) ? plan->in : plan->buf1;
            const bool  input_intact = !plan->exch && plan->wbuf && src != plan->buf2 && !(plan->flags & (DFFT_PLAN_UNFUSED | DFFT_PLAN_NATURAL));
            if (!input_intact) return rc;
            if (getenv("DFFT_DEBUG")) fprintf(stderr, "[dfft] %s -- running the transform again on the two-launch stage\n", dfft_last_error());
            rc = run();
            if (rc) return rc;
        }
    }
    if (zrc0) return fail(zrc0, zmsg0);  // (asynchronous execute of a P > 1 plan: queued for the peers' sake, invalid all the same)
    // host-synchronised executes have drained the stream: an asynchronous exchange that timed out on a dead or slow peer
    // must not let the caller print timings / use results (the reference-named wrapper always executes this way)
    if (sync && plan->comm) {
        rc = comm_check(plan->comm);
        if (rc) return rc;
    }
    if ((exec_flags & DFFT_EXEC_PRINT) && plan->direction == DFFT_FORWARD) {
        double t[4];
        rc = dfft_stage_times(plan, t);
        if (rc) return rc;
        // fft_mpi_3d_api.cpp:201
        printf("t0: %lf, t1: %lf, t2: %lf, t3: %lf, total: %lf\n", t[0], t[1], t[2], t[3], t[0] + t[1] + t[2] + t[3]);
    }
    return DFFT_OK;
}

// X-pass kernel of the plan alone on hand-over buffer `w` (forward: w -> result buffer, backward: source -> w): median of
// five timed launches after two warm-up launches, in ms.  Results are garbage (w is not initialised); only the addresses matter.
static int probe_x_pass(dfft_plan_s* p, void* w, float* ms_out) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    DFFT_HIP_TRY(hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) {
        (void)hipEventDestroy(e0);
        return fail(DFFT_EHIP, "dfft_plan_tune: cannot create events");
    }
    float t[5];
    int   rc = DFFT_OK;
    for (int i = 0; i < 7 && rc == DFFT_OK; ++i) {
        hipError_t e = hipEventRecord(e0, p->stream);
        if (e == hipSuccess) {
            if (p->direction == DFFT_FORWARD) rc = launch_x(p, w, p->buf2, false, 0, &p->wl);
            else rc = launch_x(p, (p->flags & DFFT_PLAN_INPUT_FROM_IN) ? p->in : p->buf1, w, false, 0, &p->wl);
        }
        if (e == hipSuccess && rc == DFFT_OK) e = hipEventRecord(e1, p->stream);
        if (e == hipSuccess && rc == DFFT_OK) e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess && rc == DFFT_OK) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess && rc == DFFT_OK) rc = fail(DFFT_EHIP, std::string("dfft_plan_tune: ") + hipGetErrorString(e));
        if (i >= 2) t[i - 2] = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    std::sort(t, t + 5);
    *ms_out = t[2];
    return DFFT_OK;
}

int dfft_plan_tune(dfft_plan_t plan) {
    if (!plan) return fail(DFFT_EINVAL, "dfft_plan_tune: null plan");
    const char* te = getenv("DFFT_TUNE");
    if (te && *te == '0') return DFFT_OK;
    // only fused single-GPU plans with a hand-over buffer have anything to place
    if (plan->exch || !plan->wbuf || (plan->flags & (DFFT_PLAN_NATURAL | DFFT_PLAN_UNFUSED))) return DFFT_OK;
    dfft_plan_s* p = plan;
    // plan-time call: wait for the whole device, not only for the plan's stream -- the caller may have filled `out` on a stream of
    // its own (the contents are set aside below and put back at the end; seen as a lost update in the GPU suite of round 4)
    DFFT_HIP_TRY(hipDeviceSynchronize());
    const size_t wbytes = (size_t)p->xs * p->wl.plane * elem_bytes(p->dtype);
    // Candidates: the current buffer, then fresh allocations of the same size, ALL kept until the end.  The driver hands out
    // device memory block by block (runs of 2 ... 36 consecutive 2 GiB allocations behave alike, then the behaviour flips:
    // tools/xprobe.hip census, profiles/r03/experiments/xprobe_census_*.log, placement_state.log), so every allocation that stays
    // alive moves the next one on, and walking densely cannot step over a short run the way a few large strides can (an
    // earlier version placed seven candidates behind spacers of 8 ... 64 GiB; on some boxes all of them behaved like the first
    // one: experiments/tune_check_final.log).  The walk ends at the first candidate that is 3 % away from another one (the two
    // behaviours are 5-8 % apart), after DFFT_TUNE_TRIES candidates (default 128), or when the transient footprint would exceed
    // 70 % of the free device memory; one probe costs about 7 X passes.  DFFT_TUNE_SPACER_MB puts a spacer in front of every
    // candidate (coarser, further-reaching walk).
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
        (void)hipGetLastError();
        free_b = total_b = 0;
    }
    // a process that finds at least 90 % of the device's memory free has the GPU to itself: the long walk (what bench.py asks for
    // explicitly) -- the drop-in CLI then lands in the fast mode as reliably as the benchmark does; otherwise the short, polite one
    const bool  alone = total_b > 0 && free_b >= total_b / 10 * 9;
    int         max_tries = alone ? 128 : 32;
    const char* mt = getenv("DFFT_TUNE_TRIES");
    if (mt && atoi(mt) > 0) max_tries = atoi(mt);
    size_t      spacer_bytes = 0;
    const char* sm = getenv("DFFT_TUNE_SPACER_MB");
    if (sm && atoll(sm) > 0) spacer_bytes = (size_t)atoll(sm) << 20;
    // transient footprint: at most a quarter of the free device memory by default (DFFT_TUNE_MEM_PCT, 1 ... 90) -- other plans and
    // processes share the GPU -- and 70 % when this process has the GPU to itself (above)
    int         pct = alone ? 70 : 25;
    const char* pe = getenv("DFFT_TUNE_MEM_PCT");
    if (pe && atoi(pe) >= 1 && atoi(pe) <= 90) pct = atoi(pe);
    const size_t budget = free_b / 100 * (size_t)pct;
    // The probe launches of a forward plan write (garbage) into the plan's result buffer -- the caller's `out`, which the
    // reference's plan creation never touches (fft_mpi_3d_api.cpp:41-141): its contents are set aside and put back.  Without
    // room for that copy nothing is tuned.
    void*        saved_out = nullptr;
    const size_t out_bytes = (size_t)p->max_count * elem_bytes(p->dtype);
    if (p->direction == DFFT_FORWARD) {
        if (out_bytes + wbytes > budget || hipMalloc(&saved_out, out_bytes) != hipSuccess) {
            (void)hipGetLastError();
            return DFFT_OK;
        }
        if (hipMemcpyAsync(saved_out, p->buf2, out_bytes, hipMemcpyDeviceToDevice, p->stream) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(saved_out);
            return DFFT_OK;
        }
    }
    auto restore_out = [&]() {
        if (!saved_out) return;
        (void)hipMemcpyAsync(p->buf2, saved_out, out_bytes, hipMemcpyDeviceToDevice, p->stream);
        (void)hipStreamSynchronize(p->stream);
        (void)hipFree(saved_out);
        saved_out = nullptr;
    };
    std::vector<void*> cand(1, p->wbuf), spacers;
    p->w_ms.clear();
    float ms = 0.f;
    int   rc = probe_x_pass(p, p->wbuf, &ms);
    if (rc) {
        restore_out();
        return rc;
    }
    p->w_ms.push_back(ms);
    float  lo = ms, hi = ms;
    size_t used = saved_out ? out_bytes : 0;
    while ((int)cand.size() < max_tries && used + spacer_bytes + wbytes <= budget) {
        if (lo < 0.97f * hi) {
            // Both behaviours seen?  The two classes are 5-8 % apart, but single probes of ONE class have been seen 3.5 % apart (round 4,
            // profiles/r04/experiments/tune_check_false_stop.log: 0.7841 / 0.8027 / 0.7749 ms, all slow -- the walk stopped there and the
            // bench process ran at 1.94 instead of 1.87 ms).  So the fastest and the slowest candidate are timed once more, each keeps its
            // lower figure (a probe reads high more often than low), and only a confirmed gap ends the walk.
            int ilo = 0, ihi = 0;
            for (int i = 1; i < (int)p->w_ms.size(); ++i) {
                if (p->w_ms[i] < p->w_ms[ilo]) ilo = i;
                if (p->w_ms[i] > p->w_ms[ihi]) ihi = i;
            }
            for (int idx : {ilo, ihi}) {
                float again = 0.f;
                if (probe_x_pass(p, cand[idx], &again) == DFFT_OK && again > 0.f) p->w_ms[idx] = std::min(p->w_ms[idx], again);
            }
            lo = *std::min_element(p->w_ms.begin(), p->w_ms.end());
            hi = *std::max_element(p->w_ms.begin(), p->w_ms.end());
            if (lo < 0.965f * hi) break;  // confirmed: keep the fast one
        }
        void *sp = nullptr, *nw = nullptr;
        if (spacer_bytes > 0) {
            if (hipMalloc(&sp, spacer_bytes) != hipSuccess) {
                (void)hipGetLastError();
                break;
            }
            spacers.push_back(sp);
        }
        if (slab_alloc(&nw, wbytes) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        used += spacer_bytes + wbytes;
        cand.push_back(nw);
        rc = probe_x_pass(p, nw, &ms);
        if (rc) break;
        p->w_ms.push_back(ms);
        lo = std::min(lo, ms);
        hi = std::max(hi, ms);
    }
    (void)hipStreamSynchronize(p->stream);
    for (void* sp : spacers) (void)hipFree(sp);
    // w_ms[i] belongs to candidate i (a failed probe leaves one candidate without an entry)
    int best = 0;
    for (int i = 1; i < (int)p->w_ms.size(); ++i)
        if (p->w_ms[i] < 0.985f * p->w_ms[best]) best = i;  // a later candidate must be clearly faster
    (void)hipStreamSynchronize(p->stream);
    for (int i = 0; i < (int)cand.size(); ++i)
        if (i != best) (void)slab_free(cand[i]);
    p->wbuf = cand[best];
    p->w_kept = best;
    if (rc) {
        restore_out();
        return rc;
    }
    // the kept buffer once more, now that its neighbours are gone (reported, not acted upon)
    rc = probe_x_pass(p, p->wbuf, &p->w_final_ms);
    restore_out();
    if (getenv("DFFT_DEBUG")) {
        fprintf(stderr, "[dfft] hand-over buffer placement: X pass");
        for (float v : p->w_ms) fprintf(stderr, " %.4f", v);
        fprintf(stderr, " ms, kept candidate %d (%.4f ms when re-timed)\n", p->w_kept, p->w_final_ms);
    }
    return rc;
}

int dfft_plan_describe(dfft_plan_t plan, char* buf, int len) {
    if (!plan || !buf || len < 64) return fail(DFFT_EINVAL, "dfft_plan_describe: bad arguments");
    const dfft_plan_s* p = plan;
    const bool         one = p->zy_on && !(p->flags & DFFT_PLAN_UNFUSED);
    const long long    cp = one ? zy_phase_planes(p, p->xs, p->exch && p->direction == DFFT_FORWARD) : (p->chunk_planes > 0 ? p->chunk_planes : p->xs);
    const long long    nch = cp > 0 ? (p->xs + cp - 1) / cp : 1;
    const bool         fused = !(p->flags & DFFT_PLAN_UNFUSED);
    snprintf(buf, (size_t)len,
             "pipeline=%s yz_stage=%s%s chunks=%lldx%lld handover=%s rotated_exchange_rows=%d overlap_parts=%lld ysub=%d parts_in_one_launch=%d tuned=%d x_variant=%s",
             (p->flags & DFFT_PLAN_NATURAL) ? "natural" : (fused ? "fused" : "unfused"),
             (p->zy_on && fused) ? "one-launch" : "two-launches-per-chunk", (p->zy_on && fused && p->zy_lazy) ? "-lazy" : "", nch, cp,
             (fused && p->wbuf && !p->exch) ? "padded-buffer" : "bufferDev1", p->rot_elems, p->part_planes, p->ycuts, (p->zy_on && fused && p->zy_part_done) ? 1 : 0, p->w_kept >= 0 ? 1 : 0,
             (p->x_hints & FFT_HINT_HALF_PREFETCH) ? "half-prefetch" : ((p->x_hints & FFT_HINT_EARLY_WAIT) ? "early-wait" : "default"));
    return DFFT_OK;
}

int dfft_plan_tune_report(dfft_plan_t plan, int max_n, double* ms, int* kept, double* final_ms) {
    if (!plan || max_n < 0) return fail(DFFT_EINVAL, "dfft_plan_tune_report: bad arguments");
    const int n = (int)plan->w_ms.size();
    for (int i = 0; i < n && i < max_n; ++i)
        if (ms) ms[i] = plan->w_ms[i];
    if (kept) *kept = plan->w_kept;
    if (final_ms) *final_ms = plan->w_final_ms;
    return n;
}

int dfft_plan_sync(dfft_plan_t plan) {
    if (!plan) return fail(DFFT_EINVAL, "dfft_plan_sync: null plan");
    if (plan->P > 1) trace("dfft_plan_sync enter", plan->direction, plan->flags);
    DFFT_HIP_TRY(hipStreamSynchronize(plan->stream));
    if (plan->P > 1) trace("dfft_plan_sync stream drained", plan->direction, plan->flags);
    {
        const int rc = zy_check(plan);
        if (rc) return rc;
    }
    if (plan->comm) return comm_check(plan->comm);
    return DFFT_OK;
}

int dfft_stage_times(dfft_plan_t plan, double t[4]) {
    if (!plan || !t) return fail(DFFT_EINVAL, "dfft_stage_times: bad arguments");
    DFFT_HIP_TRY(hipStreamSynchronize(plan->stream));
    {
        const int rc = zy_check(plan);
        if (rc) return rc;
    }
    if (plan->comm) {
        const int rc = comm_check(plan->comm);
        if (rc) return rc;
    }
    if (!plan->timed) return fail(DFFT_EINVAL, "dfft_stage_times: the last execute ran with DFFT_EXEC_NO_TIMING");
    if (plan->host_timed) {
        for (int i = 0; i < 4; ++i) t[i] = plan->host_t[i];
        return DFFT_OK;
    }
    for (int i = 0; i < 4; ++i) {
        float ms = 0;
        DFFT_HIP_TRY(hipEventElapsedTime(&ms, plan->ev[i], plan->ev[i + 1]));
        t[i] = ms * 1e-3;
    }
    return DFFT_OK;
}

int dfft_kernel_times(dfft_plan_t plan, double t[3]) {
    if (!plan || !t) return fail(DFFT_EINVAL, "dfft_kernel_times: bad arguments");
    if (plan->host_timed || !plan->timed)
        return fail(DFFT_EINVAL, "dfft_kernel_times: needs an execute without DFFT_EXEC_SYNC_STAGES / DFFT_EXEC_NO_TIMING");
    if (plan->flags & DFFT_PLAN_UNFUSED) return fail(DFFT_EINVAL, "dfft_kernel_times: fused plans only");
    if (plan->zy_on)
        return fail(DFFT_EINVAL, "dfft_kernel_times: not available -- this plan runs its Z and Y passes inside ONE launch (no event between them); "
                                 "create it with DFFT_T0_ONE_LAUNCH=0 to time the passes separately");
    if (plan->chunk_planes > 0 || plan->part_planes > 0)
        return fail(DFFT_EINVAL, "dfft_kernel_times: Z and Y launches are interleaved per cache chunk (set DFFT_CHUNK_MB=0)");
    DFFT_HIP_TRY(hipStreamSynchronize(plan->stream));
    float z = 0, y = 0, x = 0;
    if (plan->direction == DFFT_FORWARD) {
        DFFT_HIP_TRY(hipEventElapsedTime(&z, plan->ev[0], plan->ev[5]));
        DFFT_HIP_TRY(hipEventElapsedTime(&y, plan->ev[5], plan->ev[1]));
        DFFT_HIP_TRY(hipEventElapsedTime(&x, plan->ev[3], plan->ev[4]));
    } else {
        DFFT_HIP_TRY(hipEventElapsedTime(&x, plan->ev[0], plan->ev[1]));
        DFFT_HIP_TRY(hipEventElapsedTime(&y, plan->ev[3], plan->ev[5]));
        DFFT_HIP_TRY(hipEventElapsedTime(&z, plan->ev[5], plan->ev[4]));
    }
    t[0] = z * 1e-3;
    t[1] = y * 1e-3;
    t[2] = x * 1e-3;
    return DFFT_OK;
}

int dfft_plan_destroy(dfft_plan_t plan) {
    if (!plan) return DFFT_OK;
    if (plan->P > 1) trace("dfft_plan_destroy", plan->direction, plan->flags);
    if (plan->stream) hipStreamSynchronize(plan->stream);
    if (plan->stream2) hipStreamSynchronize(plan->stream2);
    const int zy_rc = zy_check(plan);  // an execute nobody synchronised through the library: its failure is reported here at the latest
    if (plan->comm) {  // same order as the registrations (the IPC communicator makes these collective)
        comm_unregister(plan->comm, plan->me, plan->xd2.slot);
        comm_unregister(plan->comm, plan->me, plan->xd.slot);
    }
    for (auto& e : plan->ev)
        if (e) hipEventDestroy(e);
    for (auto& e : plan->part_ev)
        if (e) hipEventDestroy(e);
    if (plan->join_ev) hipEventDestroy(plan->join_ev);
    for (auto& e : plan->y_ev)
        if (e) hipEventDestroy(e);
    if (plan->stream2) hipStreamDestroy(plan->stream2);
    if (plan->stream) hipStreamDestroy(plan->stream);
    (void)comm_recv_free(plan->comm, plan->buf1);
    (void)comm_recv_free(plan->comm, plan->rbuf);
    if (plan->wbuf) slab_free(plan->wbuf);
    if (plan->lbuf) hipFree(plan->lbuf);
    if (plan->zy_ctl) hipFree(plan->zy_ctl);
    if (plan->zy_part_done) hipFree(plan->zy_part_done);
    if (plan->zy_err) hipHostFree(plan->zy_err);
    delete plan;
    return zy_rc;
}

int dfft_fft1d_rows(void* in, void* out, long long n, long long batch, int dtype, int direction, void* stream) {
    if (!in || !out || batch < 0 || (dtype != DFFT_F64 && dtype != DFFT_F32) || (direction != DFFT_FORWARD && direction != DFFT_BACKWARD))
        return fail(DFFT_EINVAL, "dfft_fft1d_rows: bad arguments");
    if (!dfft_length_supported(n)) return fail(DFFT_EUNSUPPORTED, "dfft_fft1d_rows: unsupported length");
    if (dfft_device_count() < 1) return fail(DFFT_ENOGPU, "dfft_fft1d_rows: no HIP device visible (no CPU fallback)");
    return fft_rows(in, out, (int)n, batch, dtype, direction, (hipStream_t)stream);
}

// ---- batched 2D transform: the plan's t0 stage as an entry point of its own ------------------------------------
// State of the one-launch stage for one (device, stream, plane shape, structure): the kernel's control block, its pinned error word and
// the host's running ticket / execute counters (dfft_zy.hip) -- what a plan keeps in dfft_plan_s, cached here because the call is
// plan-less.  Freed by dfft_trim().
namespace {
struct Zy2dCtx {
    ZyCtl*    ctl = nullptr;
    unsigned* err = nullptr;
    unsigned  ticket = 0, execs = 0;
    bool      off = false;  // a launch gave up: this context stays on two launches per chunk
};
struct Zy2dKey {
    int         dev;
    hipStream_t stream;
    int         n1, n2, sign;
    bool        operator<(const Zy2dKey& o) const { return std::tie(dev, stream, n1, n2, sign) < std::tie(o.dev, o.stream, o.n1, o.n2, o.sign); }
};
std::mutex                 g_zy2d_mutex;
std::map<Zy2dKey, Zy2dCtx> g_zy2d;

void zy2d_trim() {
    std::lock_guard<std::mutex> lk(g_zy2d_mutex);
    int                         cur = 0;
    (void)hipGetDevice(&cur);
    for (auto& kv : g_zy2d) {
        (void)hipSetDevice(kv.first.dev);
        (void)hipDeviceSynchronize();
        if (kv.second.ctl) (void)hipFree(kv.second.ctl);
        if (kv.second.err) (void)hipHostFree(kv.second.err);
    }
    g_zy2d.clear();
    (void)hipSetDevice(cur);
}
}  // namespace

// `batch` planes of [n1][n2] (n2 contiguous), each transformed along both axes; in place (out == in) or out of place (`in` is left
// untouched).  What a 3D plan runs as t0 (reference fftZY, fft_mpi_3d_api.cpp:466-522; templateFFT's FFTDim = 2 application,
// templateFFT.cpp:5767): planes are taken in groups that fit the 256 MiB Infinity Cache, so the column pass reads what the row pass
// wrote from the cache, and plane shapes the one-launch stage is built for (dfft_zy.hip: fp64, n1 in {256, 512, 768 (n2 = 512)}, n2 in
// {256, 512}) run as ONE persistent launch.  Forward: rows n2 then columns n1; backward: the same order with conjugated twiddles
// (the two 1-D transforms of a plane commute).  Un-normalised.
int dfft_fft2d_batch(void* in, void* out, long long n1, long long n2, long long batch, int dtype, int direction, void* stream) {
    if (!in || !out || n1 < 1 || n2 < 1 || batch < 0 || (dtype != DFFT_F64 && dtype != DFFT_F32) || (direction != DFFT_FORWARD && direction != DFFT_BACKWARD))
        return fail(DFFT_EINVAL, "dfft_fft2d_batch: bad arguments");
    if (!dfft_length_supported(n1) || !dfft_length_supported(n2)) return fail(DFFT_EUNSUPPORTED, "dfft_fft2d_batch: unsupported length");
    if (dfft_device_count() < 1) return fail(DFFT_ENOGPU, "dfft_fft2d_batch: no HIP device visible (no CPU fallback)");
    if (batch == 0) return DFFT_OK;
    hipStream_t     s = (hipStream_t)stream;
    const long long plane = n1 * n2, plane_b = plane * (long long)elem_bytes(dtype);
    if (n1 > 4096 || n2 > 4096 || plane >= (1ll << 31)) {  // four-step axes: whole-buffer passes on the 1-D entry points
        int rc = dfft_fft1d_rows(in, out, n2, n1 * batch, dtype, direction, stream);
        if (rc == DFFT_OK) rc = dfft_fft1d_cols(out, out, n1, n2, batch, dtype, direction, stream);
        return rc;
    }
    static const bool one_launch_off = [] {
        const char* e = getenv("DFFT_T0_ONE_LAUNCH");
        return e && *e == '0';
    }();
    // ---- one persistent launch
    if (!one_launch_off && zy_supported(dtype, (int)n1, (int)n2) && batch <= ZY_MAX_PLANES) {
        int dev = 0;
        DFFT_HIP_TRY(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(g_zy2d_mutex);
        Zy2dCtx&                    c = g_zy2d[Zy2dKey{dev, s, (int)n1, (int)n2, direction}];
        if (!c.off && !c.ctl) {
            // (zeroed on the caller's stream and waited for, like a plan's: dfft_plan_create)
            if (hipMalloc((void**)&c.ctl, sizeof(ZyCtl)) != hipSuccess || hipMemsetAsync(c.ctl, 0, sizeof(ZyCtl), s) != hipSuccess ||
                hipStreamSynchronize(s) != hipSuccess || hipHostMalloc((void**)&c.err, 128, hipHostMallocMapped) != hipSuccess) {
                (void)hipGetLastError();
                if (c.ctl) (void)hipFree(c.ctl);
                c.ctl = nullptr;
                c.off = true;
            } else {
                *c.err = 0u;
            }
        }
        if (!c.off && *(volatile unsigned*)c.err != 0u) {
            c.off = true;  // sticky on the device side too: every later launch on this block would return at once
            return fail(DFFT_EHIP, "dfft_fft2d_batch: an earlier one-launch YZ stage on this stream gave up; the results of that call and of every call queued "
                                   "behind it are invalid, later calls use two launches per chunk");
        }
        if (!c.off) {
            const void *twz = nullptr, *twy = nullptr;
            DFFT_TRY(get_twiddles((int)n2, dtype, &twz));
            DFFT_TRY(get_twiddles((int)n1, dtype, &twy));
            ZyLaunch L;
            std::memset(&L, 0, sizeof(L));
            L.dtype = dtype;
            L.n1 = (int)n1;
            L.n2 = (int)n2;
            L.dir = +1;  // rows first in both directions (the inverse: SIGN = -1, dfft_zy.hip)
            L.sign = direction;
            L.src = in;
            L.w = out;
            L.dst = nullptr;
            L.src_plane = L.w_plane = L.dst_plane = plane;
            DFFT_ZY_SET_PITCH(L, n2);
            L.plane0 = 0;
            L.nplanes = batch;
            const long long fit = std::max(1ll, (230ll << 20) / plane_b), nch = (batch + fit - 1) / fit;
            L.chunk = batch * plane_b <= (256ll << 20) ? batch : (batch + nch - 1) / nch;  // zy_phase_planes
            L.ctl = c.ctl;
            L.twz = twz;
            L.twy = twy;
            L.lazy = 1;
            static const unsigned polls = [] {
                const char* sp = getenv("DFFT_ZY_SPIN_POLLS");
                return sp && atoll(sp) > 0 ? (unsigned)std::min(atoll(sp), 0xffffffffll) : (4u << 20);
            }();
            L.spin_polls = polls;
            if (hipHostGetDevicePointer((void**)&L.err_host, c.err, 0) != hipSuccess) return fail(DFFT_EHIP, "dfft_fft2d_batch: no device pointer for the error word");
            unsigned producers = 0;
            (void)zy_units_per_plane(L.n1, L.n2, L.dir, 0, &producers);
            L.ticket_base = c.ticket;
            L.done_base = c.execs * producers;
            ++c.execs;
            c.ticket += zy_tickets(L.n1, L.n2, L.dir, 0, L.nplanes, L.chunk);
            return check_launch(launch_zy(L, s), "dfft_fft2d_batch (one-launch YZ stage)");
        }
    }
    // ---- two launches per Infinity-Cache chunk (dfft_plan_create's rule for single-GPU plans)
    long long cp = batch;
    if (batch * plane_b > (256ll << 20)) {
        long long fit = std::max(1ll, (256ll << 20) / plane_b);
        if (plane_b >= (8ll << 20) && fit > 1) --fit;
        const long long nchunks = std::max(1ll, (batch + fit - 1) / fit);
        cp = (batch + nchunks - 1) / nchunks;
    }
    const void* twy = nullptr;
    DFFT_TRY(get_twiddles((int)n1, dtype, &twy));
    const bool chunked = cp < batch;
    for (long long x0 = 0; x0 < batch; x0 += cp) {
        const long long nx = std::min(cp, batch - x0);
        DFFT_TRY(fft_rows(in, out, (int)n2, nx * n1, dtype, direction, s, x0 * n1, (chunked && in != out) ? FFT_HINT_STREAM_IN : 0));
        FftLaunch L;
        std::memset(&L, 0, sizeof(L));
        L.dtype = dtype;
        L.n = (int)n1;
        L.dir = direction;
        L.cols = 1;
        L.in = out;
        L.out = out;
        L.tw = twy;
        L.imap = L.omap = plain_axis(n1, n2, 1);
        L.itile = L.otile = TileMap{plane, 1};
        L.na = nx;
        L.a_first = x0;
        L.ncols = (int)n2;
        DFFT_TRY(check_launch(launch_fft(L, s), "dfft_fft2d_batch (columns)"));
    }
    return DFFT_OK;
}

int dfft_trim(void) {
    long_scratch_trim();
    zy2d_trim();
    return DFFT_OK;
}

int dfft_scale(void* data, long long count, int dtype, double s, void* stream) {
    if (!data || count < 0 || (dtype != DFFT_F64 && dtype != DFFT_F32)) return fail(DFFT_EINVAL, "dfft_scale: bad arguments");
    if (dfft_device_count() < 1) return fail(DFFT_ENOGPU, "dfft_scale: no HIP device visible (no CPU fallback)");
    hipError_t e = launch_scale(dtype, data, count, s, (hipStream_t)stream);
    if (e != hipSuccess) return fail(DFFT_EHIP, std::string("dfft_scale: ") + hipGetErrorString(e));
    return DFFT_OK;
}

int dfft_fft1d_cols(void* in, void* out, long long n, long long width, long long batch, int dtype, int direction,
                    void* stream) {
    if (!in || !out || batch < 0 || width < 1 || (dtype != DFFT_F64 && dtype != DFFT_F32) ||
        (direction != DFFT_FORWARD && direction != DFFT_BACKWARD))
        return fail(DFFT_EINVAL, "dfft_fft1d_cols: bad arguments");
    if (!dfft_length_supported(n)) return fail(DFFT_EUNSUPPORTED, "dfft_fft1d_cols: unsupported length");
    if (dfft_device_count() < 1) return fail(DFFT_ENOGPU, "dfft_fft1d_cols: no HIP device visible (no CPU fallback)");
    if (n > 4096) {
        if (batch == 0) return DFFT_OK;
        LongScratchLease lease = nullptr;
        void*            scr = long_scratch((size_t)batch * n * width * elem_bytes(dtype), (hipStream_t)stream, &lease);
        if (!scr) return fail(DFFT_EHIP, "dfft_fft1d_cols: cannot allocate the scratch buffer of the four-step transform");
        const int rc = long_fft(in, out, n, width, batch, dtype, direction, 1.0, scr, (hipStream_t)stream);
        long_scratch_release(lease);
        return rc;
    }
    const void* tw = nullptr;
    int         rc = get_twiddles((int)n, dtype, &tw);
    if (rc) return rc;
    FftLaunch L;
    std::memset(&L, 0, sizeof(L));
    L.dtype = dtype;
    L.n = (int)n;
    L.dir = direction;
    L.cols = 1;
    L.in = in;
    L.out = out;
    L.tw = tw;
    L.imap = L.omap = plain_axis(n, width, 1);
    L.itile = L.otile = TileMap{n * width, 1};
    L.na = batch;
    L.ncols = (int)width;
    return check_launch(launch_fft(L, (hipStream_t)stream), "dfft_fft1d_cols");
}

}  // extern "C"
