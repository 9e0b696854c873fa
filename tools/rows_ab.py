"""Developer A/B (GPU box): in-place batched rows of the long single-pass lengths, 1 GiB of fp64 / 512 MiB of fp32 like Test_1D.
DFFT_LIB selects the library build (tools/build_variant.py)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from distributedfft_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
for dt, code, S in ((torch.complex128, 0, 16), (torch.complex64, 1, 8)):
    for n in (2048, 4096, 2401, 3125, 2187):
        batch = (1 << 26) // n
        x = torch.rand(batch, n, dtype=torch.float64, device=dev).to(dt)
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            L.check(lib.dfft_fft1d_rows(x.data_ptr(), x.data_ptr(), n, batch, code, 1, s), "rows")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            L.check(lib.dfft_fft1d_rows(x.data_ptr(), x.data_ptr(), n, batch, code, 1, s), "rows")
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        import math
        print("rows n=%4d %s: %.4f ms  %.0f GB/s  %.0f GFlop/s" % (n, "f64" if code == 0 else "f32", ms, 2 * S * batch * n / ms / 1e6,
                                                                 5 * batch * n * math.log2(n) / ms / 1e6), flush=True)
