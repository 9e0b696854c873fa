"""Developer probe (GPU box, run under rocprofv3 --pmc): the kernels of the two BASELINE configurations that sit furthest from the
roofline -- config 4 (1024 x 768 x 512 fp64) and config 5 (2048 x 2048 x 1024 fp32), rank 0's local work of the P = 8 decomposition with
the exchange switched off -- next to the 512^3 fp64 single-GPU plan (the headline's kernels, as the control) and the stand-alone
1024- / 2048-point column and row kernels of tools/long_axis_bench.py.  A few launches of each, so that per-kernel counters can be
read side by side (a limiter table for the long-axis kernels).   usage: long_axis_probe.py [executes]"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["DFFT_EXCHANGE_NOOP"] = "1"  # read once per process: P > 1 plans run rank 0's local work only
from distributedfft_amd import _lib as L  # noqa: E402
from distributedfft_amd import api  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
lib = L.load()

for size, prec, P in (((512, 512, 512), "fp64", 1), ((1024, 768, 512), "fp64", 8), ((2048, 2048, 1024), "fp32", 8)):
    cdt = torch.complex128 if prec == "fp64" else torch.complex64
    mc = api.get_max_data_count(*size, P, False)
    a = (torch.rand(mc, device=dev, dtype=torch.float32) - 0.5).to(cdt)
    b = torch.zeros_like(a)
    comm = api.Comm.local(P) if P > 1 else None
    p = api.Plan(*size, a, b, comm, 0, P, api.FORWARD, api.PLAN_INPUT_FROM_IN)
    for _ in range(reps):
        p.execute(api.EXEC_NO_TIMING)
    p.execute()
    p.sync()
    print("plan", size, prec, "P", P, "stage times ms", [round(1e3 * t, 4) for t in p.stage_times()], p.describe(), flush=True)
    p.destroy()
    if comm:
        comm.destroy()
    del a, b
    torch.cuda.empty_cache()

s = torch.cuda.current_stream().cuda_stream
total = 1 << 26
for dtype, code in ((torch.complex128, 0), (torch.complex64, 1)):
    for n in (512, 1024, 2048):
        x = (torch.rand(total, dtype=torch.float32, device=dev) - 0.5).to(dtype)
        y = torch.empty_like(x)
        for _ in range(reps):
            lib.dfft_fft1d_rows(x.data_ptr(), y.data_ptr(), n, total // n, code, 1, s)
            lib.dfft_fft1d_cols(x.data_ptr(), y.data_ptr(), n, 512, total // (n * 512), code, 1, s)
        torch.cuda.synchronize()
        del x, y
print("done", flush=True)
