"""Developer probe (GPU): t0 / t3 of single-GPU plans as a function of the Z+Y cache-chunk size (DFFT_CHUNK_PLANES), several
shapes in one process.  'auto' = the rule of dfft_plan_create.  No placement tuning (t3 carries the allocation lottery; t0 is
the quantity of interest)."""
import os, sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["DFFT_TUNE"] = "0"
from distributedfft_amd import api

dev = torch.device("cuda:0")
CASES = [
    ((512, 512, 512), torch.complex128, ["auto", 64, 72, 80, 96, 128, 256, 512, "auto"]),
    ((512, 512, 512), torch.complex64, ["auto", 64, 96, 103, 128, 192, 256]),
    ((384, 384, 384), torch.complex128, ["auto", 64, 96, 113, 128, 192]),
    ((768, 768, 768), torch.complex128, ["auto", 16, 24, 28, 32]),
    ((1024, 768, 512), torch.complex128, ["auto", 32, 41, 42, 64]),
    ((1024, 1024, 1024), torch.complex128, ["auto", 8, 15, 16, 32]),
    ((2048, 1024, 512), torch.complex128, ["auto", 16, 31, 32]),
]
for N, dt, settings in CASES:
    n = N[0] * N[1] * N[2]
    rdt = torch.float64 if dt == torch.complex128 else torch.float32
    a = torch.complex(torch.rand(n, device=dev, dtype=rdt) * 2 - 1, torch.rand(n, device=dev, dtype=rdt) * 2 - 1)
    b = torch.zeros_like(a)
    for cp in settings:
        if cp == "auto":
            os.environ.pop("DFFT_CHUNK_PLANES", None)
        else:
            os.environ["DFFT_CHUNK_PLANES"] = str(cp)
        p = api.Plan(*N, a, b, None, 0, 1, api.FORWARD, api.PLAN_INPUT_FROM_IN)
        for _ in range(3):
            p.execute(api.EXEC_NO_TIMING)
        ts = []
        for _ in range(9):
            p.execute(api.EXEC_ASYNC)
            ts.append(p.stage_times())
        ts = np.median(np.array(ts), axis=0) * 1e3
        print(f"{'x'.join(map(str, N)):>14} {'f64' if dt == torch.complex128 else 'f32'} chunk {str(cp):>5}: t0 {ts[0]:8.4f}  t3 {ts[3]:8.4f}  total {ts.sum():8.4f} ms", flush=True)
        p.destroy()
    del a, b
    torch.cuda.empty_cache()
