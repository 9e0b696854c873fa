"""Developer probe (GPU box, run under rocprofv3 --pmc): one forward and one backward single-GPU plan of the same size, a few executes each,
so that the counters of the one-launch YZ stage's forward (DIR = 1) and inverse (DIR = -1) kernels can be read side by side
(why is the inverse YZ stage of 512^3 fp64 20 % slower?).   usage: fwd_bwd_probe.py [n0xn1xn2] [executes]"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from distributedfft_amd import api  # noqa: E402

size = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "512x512x512").split("x"))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
mc = api.get_max_data_count(*size, 1, True)
a = (torch.rand(mc, device=dev, dtype=torch.float64) - 0.5).to(torch.complex128)
b, c = torch.zeros_like(a), torch.zeros_like(a)
for direction, src, dst in ((api.FORWARD, a, b), (api.BACKWARD, b, c)):
    p = api.Plan(*size, src, dst, None, 0, 1, direction, api.PLAN_INPUT_FROM_IN)
    for _ in range(reps):
        p.execute(api.EXEC_NO_TIMING)
    p.execute()
    p.sync()
    print("direction", direction, "stage times ms", [round(1e3 * t, 4) for t in p.stage_times()], p.describe(), flush=True)
    p.destroy()
