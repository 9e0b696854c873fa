"""Developer micro-benchmark (GPU box): per-stage times and HBM GB/s of the single-GPU pipeline. Not the graded bench."""
import math
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from distributedfft_amd import api  # noqa: E402


def bench_plan(N, dtype=torch.complex128, flags=api.PLAN_INPUT_FROM_IN, reps=10):
    dev = torch.device("cuda:0")
    n = N[0] * N[1] * N[2]
    a = (torch.rand(n, dtype=torch.float64, device=dev) * 2 - 1).to(dtype) + 1j * (torch.rand(n, dtype=torch.float64, device=dev) * 2 - 1).to(dtype)
    b = torch.zeros_like(a)
    torch.cuda.synchronize()
    plan = api.Plan(*N, a, b, None, 0, 1, api.FORWARD, flags)
    for _ in range(3):
        plan.execute()
    plan.sync()
    ts = []
    for _ in range(reps):
        plan.execute()
        ts.append(plan.stage_times())
    ts = np.array(ts)
    med = np.median(ts, axis=0)
    tot = float(np.median(ts.sum(axis=1)))
    S = 16 if dtype == torch.complex128 else 8
    gf = 5.0 * n * math.log2(n) * 1e-9 / tot
    print(f"N={N} {str(dtype).split('.')[-1]} flags={flags}: t0={med[0]*1e3:.3f} t1={med[1]*1e3:.3f} t2={med[2]*1e3:.3f} "
          f"t3={med[3]*1e3:.3f} ms total={tot*1e3:.3f} ms  {gf:.0f} GFlops/s | t0 {4*S*n/med[0]/1e9:.0f} GB/s (2 passes) "
          f"t3 {2*S*n/med[3]/1e9:.0f} GB/s | local 4S·N/t = {4*S*n/tot/1e9:.0f} GB/s", flush=True)
    plan.destroy()


def bench_1d(n, batch, cols=False, width=None, dtype=torch.complex128, reps=20):
    dev = torch.device("cuda:0")
    if cols:
        x = torch.rand(batch, n, width, dtype=torch.float64, device=dev).to(dtype)
    else:
        x = torch.rand(batch, n, dtype=torch.float64, device=dev).to(dtype)
    y = torch.empty_like(x)
    fn = api.fft1d_cols if cols else api.fft1d_rows
    from distributedfft_amd import _lib as L
    lib = L.load()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s = torch.cuda.current_stream().cuda_stream
    def launch():
        if cols:
            lib.dfft_fft1d_cols(x.data_ptr(), y.data_ptr(), n, width, batch, 0 if dtype == torch.complex128 else 1, 1, s)
        else:
            lib.dfft_fft1d_rows(x.data_ptr(), y.data_ptr(), n, batch, 0 if dtype == torch.complex128 else 1, 1, s)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    S = 16 if dtype == torch.complex128 else 8
    bytes_ = 2 * S * x.numel()
    print(f"{'cols' if cols else 'rows'} n={n} batch={batch} width={width} {str(dtype).split('.')[-1]}: {ms:.3f} ms  {bytes_/ms/1e6:.0f} GB/s", flush=True)


if __name__ == "__main__":
    bench_1d(512, 262144)
    bench_1d(512, 512, cols=True, width=512)
    bench_1d(256, 65536 * 4)
    bench_1d(256, 1024, cols=True, width=256)
    bench_1d(512, 262144, dtype=torch.complex64)
    bench_1d(512, 512, cols=True, width=512, dtype=torch.complex64)
    bench_1d(1024, 131072)
    bench_1d(768, 131072)
    for N in [(256, 256, 256), (512, 512, 512)]:
        bench_plan(N)
        bench_plan(N, flags=api.PLAN_INPUT_FROM_IN | api.PLAN_UNFUSED)
