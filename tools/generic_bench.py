"""Developer measurement (GPU box): rate of the run-time-scheduled kernel (lengths without a tuned plan) -- rows, columns
and a few 3D shapes.  CSV to stdout."""
import math
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from distributedfft_amd import _lib as L  # noqa: E402
from distributedfft_amd import api  # noqa: E402
from sweep_bench import time_launch, DEV  # noqa: E402

lib = L.load()
s = torch.cuda.current_stream().cuda_stream
print("kind,n,dtype,ms,GBps")
for dtype, code, S in ((torch.complex128, 0, 16), (torch.complex64, 1, 8)):
    for n in (20, 80, 200, 320, 640, 1000, 1280, 1536, 2000, 3072, 4096):
        total = 1 << 26
        batch = total // n
        x = torch.rand(batch * n, dtype=torch.float64, device=DEV).to(dtype)
        y = torch.empty_like(x)
        ms = time_launch(lambda: lib.dfft_fft1d_rows(x.data_ptr(), y.data_ptr(), n, batch, code, 1, s))
        print(f"rows,{n},{'f64' if code == 0 else 'f32'},{ms:.4f},{2 * S * batch * n / ms / 1e6:.0f}", flush=True)
        width = 512
        b2 = max(1, total // (n * width))
        ms = time_launch(lambda: lib.dfft_fft1d_cols(x.data_ptr(), y.data_ptr(), n, width, b2, code, 1, s))
        print(f"cols,{n},{'f64' if code == 0 else 'f32'},{ms:.4f},{2 * S * b2 * n * width / ms / 1e6:.0f}", flush=True)
print("shape,dtype,t0_ms,t3_ms,total_ms,GFlops")
for dtype, S in ((torch.complex128, 16), (torch.complex64, 8)):
    for N in ((640, 640, 640), (1000, 1000, 1000), (320, 320, 320), (1536, 512, 512)):
        n = N[0] * N[1] * N[2]
        a = torch.rand(n, dtype=torch.float64 if S == 16 else torch.float32, device=DEV).to(dtype)
        b = torch.zeros_like(a)
        plan = api.Plan(*N, a, b, None, 0, 1, api.FORWARD, api.PLAN_INPUT_FROM_IN)
        for _ in range(2):
            plan.execute()
        plan.sync()
        ts = []
        for _ in range(5):
            plan.execute()
            ts.append(plan.stage_times())
        ts = np.array(ts)
        med = np.median(ts, axis=0)
        tot = float(np.median(ts.sum(axis=1)))
        print(f"{N[0]}x{N[1]}x{N[2]},{'f64' if S == 16 else 'f32'},{med[0] * 1e3:.4f},{med[3] * 1e3:.4f},{tot * 1e3:.4f},"
              f"{5.0 * n * math.log2(n) * 1e-9 / tot:.0f}", flush=True)
        plan.destroy()
        del a, b
        torch.cuda.empty_cache()
