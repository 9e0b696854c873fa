"""Developer probe (GPU): how do the stage times of the 512^3 fp64 single-GPU plan depend on where the padded hand-over buffer
starts INSIDE one allocation?  DFFT_W_SLACK_MB makes the plan allocate slack behind the buffer, DFFT_W_OFFSET (read at every
execute) moves the data start.  Same physical pages, same code, only the offset changes.  Two plans (two allocations)."""
import os, sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["DFFT_W_SLACK_MB"] = "640"
os.environ["DFFT_TUNE"] = "0"
from distributedfft_amd import api

dev = torch.device("cuda:0")
n = 512
a = (torch.rand(n ** 3, device=dev, dtype=torch.float64) * 2 - 1).to(torch.complex128)
b = torch.zeros_like(a)
MiB = 1 << 20
offs = [0, 128, 4096, 65536, MiB, 2 * MiB, 3 * MiB, 4 * MiB, 6 * MiB, 8 * MiB, 12 * MiB, 16 * MiB, 24 * MiB, 32 * MiB, 48 * MiB,
        64 * MiB, 96 * MiB, 128 * MiB, 192 * MiB, 256 * MiB, 384 * MiB, 512 * MiB, 0]


def times(plan):
    for _ in range(3):
        plan.execute(api.EXEC_NO_TIMING)
    ts = []
    for _ in range(7):
        plan.execute(api.EXEC_ASYNC)
        ts.append(plan.stage_times())
    ts = np.median(np.array(ts), axis=0)
    return ts[0] * 1e3, ts[3] * 1e3


keep = []
for rep in range(3):
    p = api.Plan(n, n, n, a, b, None, 0, 1, api.FORWARD, api.PLAN_INPUT_FROM_IN)
    print(f"plan {rep}: in {a.data_ptr():x} out {b.data_ptr():x}", flush=True)
    for off in offs:
        os.environ["DFFT_W_OFFSET"] = str(off)
        t0, t3 = times(p)
        print(f"  offset {off / MiB:10.4f} MiB: t0 {t0:.4f}  t3 {t3:.4f} ms", flush=True)
    os.environ["DFFT_W_OFFSET"] = "0"
    keep.append(p)  # keep the allocation so that the next plan lands elsewhere
for p in keep:
    p.destroy()
