"""Developer probe (GPU): X-pass time of the 512^3 fp64 plan against the plane padding of the hand-over buffer (DFFT_PAD_PLANE,
in cache lines), several plans per setting, one process, no placement tuning."""
import os, sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["DFFT_TUNE"] = "0"
from distributedfft_amd import api

dev = torch.device("cuda:0")
n = 512
a = torch.complex(torch.rand(n ** 3, device=dev, dtype=torch.float64), torch.rand(n ** 3, device=dev, dtype=torch.float64))
b = torch.zeros_like(a)
for rep in range(2):
    for pad in (3, 1, 5, 7, 9, 11, 13, 17, 21, 33, 65, 3):
        os.environ["DFFT_PAD_PLANE"] = str(pad)
        p = api.Plan(n, n, n, a, b, None, 0, 1, api.FORWARD, api.PLAN_INPUT_FROM_IN)
        for _ in range(3):
            p.execute(api.EXEC_NO_TIMING)
        ts = []
        for _ in range(9):
            p.execute(api.EXEC_ASYNC)
            ts.append(p.stage_times())
        ts = np.median(np.array(ts), axis=0) * 1e3
        print(f"rep {rep} plane pad {pad:3d} lines: t0 {ts[0]:.4f}  t3 {ts[3]:.4f} ms", flush=True)
        p.destroy()
