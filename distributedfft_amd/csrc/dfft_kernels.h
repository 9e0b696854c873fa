// dfft_kernels.h -- host-side launch descriptors for the gfx950 FFT / reorder kernels.
// Internal header (not part of the C-ABI; that is include/dfft.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dfft {

enum DType { F64 = 0, F32 = 1 };

// Maps (FFT index idx in [0,N), column c in [0,CB)) to an element offset relative to the tile base:
//   off = (idx / blk) * blk_stride + (idx % blk) * stride + c * cstride
//         + (idx / blk == nblk-1 ? a * last_delta : 0)          (uneven last slab)
// blk == N (one block) for plain strided access.  Two-level blocks (sub > 1): block ib = idx / blk splits into
// (ib / sub, ib % sub) and the block term becomes (ib / sub) * blk_stride + (ib % sub) * sub_stride -- the packed send
// layout [k][dst][x][y][N2] of the t2/t3 overlap, where k = Y sub-block within destination dst.
struct AxisMap {
    int       blk;
    int       nblk;
    long long blk_stride;
    long long stride;
    long long cstride;
    long long last_delta;
    int       sub;         // <= 1: single level
    long long sub_stride;
};

// tile -> (a = tile / tiles_per_a, b = tile % tiles_per_a); base = a * a_stride + b * CB * b_stride
struct TileMap {
    long long a_stride;
    long long b_stride;
};

// Row rotation inside the exchange buffers (P > 1 plans): row (plane x, y) of a packed / received slab is stored rotated by
// rot * x elements inside its N2 elements.  With power-of-two extents the X pass's tile is N0 segments of 128 bytes exactly
// ys * N2 * S bytes apart (1 MiB at 512^3 fp64 / P = 4), which all fall on the same memory channels; the rotation moves
// consecutive planes' segments to different lines at no cost in bytes (the single-GPU plan pads its own hand-over buffer
// instead).  The Y pass applies it per tile (the plane is the tile's slow index), the X pass per FFT point (the plane is the
// FFT index).  Messages keep their sizes and offsets, only the order of the elements inside a row changes, and only between
// the two passes of this library -- no caller-visible buffer is affected.
struct RotMap {
    int in_mode, out_mode;  // 0: none; 1: by the tile's plane index a + a0; 2: by the FFT point index
    int rot;                // elements (units of one V) per plane
    int mask;               // row length - 1 in units of V (row length is a power of two)
    int a0;                 // global index of the launch's plane 0 (mode 1)
};

struct FftLaunch {
    int         dtype;  // DType
    int         n;      // FFT length
    int         dir;    // +1 forward, -1 backward (unnormalised both ways)
    int         cols;   // 0: "row" kernel (one FFT per tile, contiguous, wave-local);
                        // 1: "column" kernel (CB adjacent columns per tile, block-wide)
    const void* in;
    void*       out;
    const void* tw;     // N-entry table of e^{-2 pi i k / N} in the kernel's dtype
    AxisMap     imap, omap;
    TileMap     itile, otile;
    long long   na;      // column launches: number of `a` slices, each ncols columns wide (tile counts are derived)
    long long   ntiles;  // row launches: number of FFTs.  (Column launches: filled in by the variant selection.)
    int         tiles_per_a;
    int         ncols;  // number of valid columns along the tiled dimension (guard for ragged last tile)
    long long   a_first;  // first `a` this launch covers (plane-chunked launches); ntiles counts tiles from there
    int         hints;    // FFT_HINT_* cache-policy hints (never change results)
    double      scale;    // results are multiplied by this before the store (0 or 1 = no scaling)
    int         blocks_per_cu_limit;  // > 0: cap the persistent grid at this many blocks per CU (leaves room for a
                                      // kernel running concurrently on another stream)
    RotMap      rot;                  // row rotation of an exchange-buffer side (all zero: none)
    int         grid_limit;           // > 0: cap the persistent grid at this many workgroups (HBM writes run faster from
                                      // fewer concurrent writers, profiles/r02/README.md section 1; tuning knob)
};
enum {
    FFT_HINT_STREAM_IN = 1,   // input is read once and must not displace the cache-resident chunk: non-temporal loads
    FFT_HINT_STREAM_OUT = 2,  // output goes to a different buffer and is not re-read soon: non-temporal stores
    // kernel-variant switches of the staged transposing store (forward X pass); measurement switches, never change results
    FFT_HINT_HALF_PREFETCH = 4,  // 16 points per thread: half-tile prefetch instead of the whole-tile one
    FFT_HINT_EARLY_WAIT = 8,     // wait for the prefetched tile before this tile's stores are issued
};

bool fft_length_supported(int n);
bool fft_length_tuned(int n);  // served by a static plan of dfft_plans.h (not by the run-time-scheduled kernel)
// run-time-scheduled kernel for 7-smooth lengths <= 4096 that have no tuned plan (dfft_generic.hip)
bool       generic_length_supported(int n);
hipError_t launch_generic(const struct FftLaunch& L, hipStream_t stream);
// Returns hipErrorInvalidValue for unsupported (n, dtype); otherwise the launch status.
hipError_t launch_fft(const FftLaunch& L, hipStream_t stream);

// Reference-structure (un-fused) reorder kernels, kept so t1 / the t3 transpose stay separately
// measurable.
//   pack  : [xl][N1][N2] -> [d][xl][yl_d][N2]   (kernel_func.cpp:73-86), dir=+1; inverse for dir=-1
//   trans : 2D transpose of a rows x cols matrix of complex elements through a padded LDS tile
//           (replaces dev_transpose_201_ept1 / _120_ept1, kernels_201.cpp:28-70, kernels_120.cpp:28-70)
hipError_t launch_pack(int dtype, int dir, const void* in, void* out, int xl, int n1, int n2, int yl,
                       int ylast, int P, hipStream_t stream);
hipError_t launch_transpose(int dtype, const void* in, void* out, long long rows, long long cols,
                            hipStream_t stream);
// out[i] = in[i] * s (used only by the optional normalised inverse of the host mirror)
hipError_t launch_scale(int dtype, void* data, long long count, double s, hipStream_t stream);

}  // namespace dfft
