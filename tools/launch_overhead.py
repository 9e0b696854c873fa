"""Developer measurement (GPU box): wall time per execute of back-to-back forward transforms with and without the
stage-boundary events (DFFT_EXEC_NO_TIMING) -- the launch-bound regime of small transforms."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from distributedfft_amd import api  # noqa: E402

dev = torch.device("cuda:0")
print("shape,dtype,us_per_execute_timed,us_per_execute_no_timing")
for dtype in (torch.complex128, torch.complex64):
    for n in (32, 64, 96, 128, 192, 256, 512):
        a = torch.rand(n ** 3, dtype=torch.float64, device=dev).to(dtype)
        b = torch.zeros_like(a)
        plan = api.Plan(n, n, n, a, b, None, 0, 1, api.FORWARD, api.PLAN_INPUT_FROM_IN)
        res = []
        for flags in (api.EXEC_ASYNC, api.EXEC_NO_TIMING):
            reps = 400 if n <= 256 else 40
            for _ in range(20):
                plan.execute(flags)
            plan.sync()
            best = 1e9
            for _ in range(3):
                t = time.perf_counter()
                for _ in range(reps):
                    plan.execute(flags)
                plan.sync()
                best = min(best, (time.perf_counter() - t) / reps)
            res.append(best * 1e6)
        print(f"{n}^3,{'f64' if dtype == torch.complex128 else 'f32'},{res[0]:.1f},{res[1]:.1f}", flush=True)
        plan.destroy()
