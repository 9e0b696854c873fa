// dfft_long.h -- axes longer than the single-pass range: four-step decomposition on the single-pass kernels (dfft_long.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace dfft {

// n = n1 * n2 with both factors tuned single-pass lengths (4096 < n <= 2^24); false if there is no such pair
bool long_split(long long n, int* n1, int* n2);
// Length-n transforms along the middle axis of data[batch][n][s] (s contiguous columns; s = 1: contiguous rows), unnormalised,
// times `scale`.  in == out is allowed.  `scratch` must hold batch * n * s elements and must not alias in / out.
int long_fft(const void* in, void* out, long long n, long long s, long long batch, int dtype, int dir, double scale, void* scratch,
             hipStream_t stream);
// scratch for callers without a plan (dfft_fft1d_rows / dfft_fft1d_cols): a grow-only buffer per (device, stream), leased to one
// host thread at a time -- long_scratch() takes the lease (nullptr buffer: nothing leased), long_scratch_release() returns it after
// the work has been enqueued.  The lease is self-contained: releasing it cannot fail and does not depend on the current device.
typedef void* LongScratchLease;
void* long_scratch(size_t bytes, hipStream_t stream, LongScratchLease* lease);
void  long_scratch_release(LongScratchLease lease);
void  long_scratch_trim();  // frees every cached buffer that is not leased (dfft_trim); drains the owning devices first

}  // namespace dfft
