"""Developer measurement (GPU box): per-rank LOCAL work of the 512^3 fp64 problem (or --size) at P ranks, measured on one GPU.
Rank 0's plan of a P-rank decomposition is executed alone with DFFT_EXCHANGE_NOOP=1 (the exchange moves nothing, results are
garbage), so t0 (Z + Y passes, Y storing the packed send layout) and t3 (X pass reading the [N0][ys][N2] receive layout) run
with the real P > 1 address maps.  Every configuration is measured with the rows of the exchange buffers rotated (DFFT_ROT=1)
and not (DFFT_ROT=0), alternating, on `reps` freshly created plans each (a plan's buffers land in different physical regions
from one creation to the next, which is worth 5-8 % of the X pass: profiles/r03/README.md section 1).
usage: local_by_P.py [n0xn1xn2] [fp64|fp32] [reps] [P,P,...] [serial|overlap]"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["DFFT_EXCHANGE_NOOP"] = "1"
from distributedfft_amd import api  # noqa: E402

size = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "512x512x512").split("x"))
prec = sys.argv[2] if len(sys.argv) > 2 else "fp64"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
cdt = torch.complex128 if prec == "fp64" else torch.complex64
S = 16 if prec == "fp64" else 8
dev = torch.device("cuda:0")
n0, n1, n2 = size
print(f"# local work per rank, {n0}x{n1}x{n2} {prec}: P, pipeline, rot, t0 ms, t3 ms (median over {reps} plans; min..max), X pass GB/s = 2 S N/P / t3")
PS = tuple(int(v) for v in sys.argv[4].split(",")) if len(sys.argv) > 4 else (1, 2, 4, 8)
for P in PS:
    if n0 % P or n1 % P:
        continue
    mc = api.get_max_data_count(n0, n1, n2, P, False)
    if mc >= 2 ** 31:  # the reference's own limit (32-bit element counts per device); the plan would be refused
        print(f"P={P} skipped: {mc} elements per device")
        continue
    a = (torch.rand(mc, device=dev, dtype=torch.float64) - 0.5).to(cdt)
    b = torch.zeros_like(a)
    comm = api.Comm.local(P) if P > 1 else None
    for flags, name in ((api.PLAN_INPUT_FROM_IN, "serial"), (api.PLAN_INPUT_FROM_IN | api.PLAN_OVERLAP, "overlap")):
        if (P == 1 and name == "overlap") or (len(sys.argv) > 5 and sys.argv[5] != name):
            continue
        res = {0: [], 1: []}
        for r in range(reps):
            for rot in (0, 1):
                os.environ["DFFT_ROT"] = str(rot)
                p = api.Plan(n0, n1, n2, a, b, comm, 0, P, api.FORWARD, flags)
                for _ in range(6):
                    p.execute(api.EXEC_NO_TIMING)
                ts = []
                for _ in range(7):
                    p.execute()
                    ts.append(p.stage_times())
                p.destroy()
                res[rot].append(np.median(np.array(ts), axis=0) * 1e3)
        for rot in (0, 1):
            m = np.array(res[rot])
            t0, t3 = np.median(m[:, 0]), np.median(m[:, 3])
            print(f"P={P} {name:8s} rot={rot}  t0 {t0:.4f} ({m[:, 0].min():.4f}..{m[:, 0].max():.4f})  t3 {t3:.4f} ({m[:, 3].min():.4f}..{m[:, 3].max():.4f})"
                  f"  X pass {2 * S * n0 * n1 * n2 / P / t3 / 1e6:.0f} GB/s", flush=True)
    if comm:
        comm.destroy()
    del a, b
    torch.cuda.empty_cache()
