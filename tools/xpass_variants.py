"""Developer A/B (GPU box): t0 / t3 of single-GPU plans with long X or Y axes under the kernel-selection switches of the process
environment (DFFT_X_DIF2, DFFT_DIF2_MIN, DFFT_NO_DUAL, DFFT_NO_DIF2 ...).  Every plan is placed with dfft_plan_tune first (the
buffer-pair lottery is worth 5-8 % of the X pass); prints the tuning report next to the times."""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from distributedfft_amd import api  # noqa: E402

dev = torch.device("cuda:0")
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(2048, 1024, 512), (1024, 768, 512), (1024, 1024, 1024), (512, 2048, 512)]
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("DFFT_") and k != "DFFT_LIB")
for N in shapes:
    n = N[0] * N[1] * N[2]
    for dt, S, name in ((torch.complex128, 16, "f64"), (torch.complex64, 8, "f32")):
        a = (torch.rand(n, device=dev, dtype=torch.float32) - 0.5).to(dt)
        b = torch.zeros_like(a)
        p = api.Plan(*N, a, b, None, 0, 1, api.FORWARD, api.PLAN_INPUT_FROM_IN)
        p.tune()
        rep = p.tune_report()
        for _ in range(4):
            p.execute(api.EXEC_NO_TIMING)
        ts = []
        for _ in range(7):
            p.execute()
            ts.append(p.stage_times())
        m = np.median(np.array(ts), axis=0) * 1e3
        print(f"[{tag}] {N[0]}x{N[1]}x{N[2]} {name}: t0 {m[0]:.3f} ms ({4 * S * n / m[0] / 1e6:.0f} GB/s for 2 passes)  t3 {m[3]:.3f} ms ({2 * S * n / m[3] / 1e6:.0f} GB/s)"
              f"  total {m.sum():.3f}  tune {rep['candidates_ms']}", flush=True)
        p.destroy()
        del a, b
        torch.cuda.empty_cache()
