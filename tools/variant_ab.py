"""Developer A/B (GPU box): kernel-variant switches that are read when a plan is created (DFFT_X_VARIANT, DFFT_ZY_LAZY, ...),
compared inside ONE process on freshly created plans, the variants interleaved `reps` times (a plan's buffers land in different
physical regions from one creation to the next, worth 5-8 % of the X pass -- profiles/r03/README.md section 1 -- so single-GPU
plans are placed with dfft_plan_tune first and every variant is measured on several plans).

usage: variant_ab.py SPEC [SPEC ...]      SPEC = n0xn1xn2:prec:P:reps:name=ENV=val+ENV=val,name=...,...
  e.g. variant_ab.py "512x512x512:fp64:1:3:base=,early=DFFT_X_VARIANT=early,lazy=DFFT_ZY_LAZY=1"
P > 1 measures rank 0's local work of a P-rank decomposition with the exchange switched off (DFFT_EXCHANGE_NOOP=1, results are
garbage), like tools/local_by_P.py.  Prints per variant the median over the plans (min..max) of t0 and t3 (each the median of 9
timed executes) and of the un-timed back-to-back rate (30 executes queued back to back, host clock around the final sync)."""
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
specs = sys.argv[1:]
if any(int(s.split(":")[2]) > 1 for s in specs):
    os.environ["DFFT_EXCHANGE_NOOP"] = "1"  # read once per process
from distributedfft_amd import api  # noqa: E402

dev = torch.device("cuda:0")
for spec in specs:
    size_s, prec, P_s, reps_s, vars_s = spec.split(":", 4)
    n0, n1, n2 = (int(v) for v in size_s.split("x"))
    P, reps = int(P_s), int(reps_s)
    variants = []
    for v in vars_s.split(","):
        name, _, envs = v.partition("=")
        variants.append((name, dict(e.split("=", 1) for e in envs.split("+") if e)))
    touched = sorted({k for _, e in variants for k in e})
    cdt = torch.complex128 if prec == "fp64" else torch.complex64
    S = 16 if prec == "fp64" else 8
    mc = api.get_max_data_count(n0, n1, n2, P, False)
    a = (torch.rand(mc, device=dev, dtype=torch.float32) - 0.5).to(cdt)
    b = torch.zeros_like(a)
    comm = api.Comm.local(P) if P > 1 else None
    res = {name: [] for name, _ in variants}
    desc = {}
    for r in range(reps):
        for name, env in variants:
            for k in touched:
                os.environ.pop(k, None)
            os.environ.update(env)
            p = api.Plan(n0, n1, n2, a, b, comm, 0, P, api.FORWARD, api.PLAN_INPUT_FROM_IN)
            if P == 1:
                p.tune()
            desc[name] = p.describe()
            for _ in range(6):
                p.execute(api.EXEC_NO_TIMING)
            ts = []
            for _ in range(9):
                p.execute()
                ts.append(p.stage_times())
            m = np.median(np.array(ts), axis=0) * 1e3
            K = 30
            p.sync()
            t_host = time.perf_counter()
            for _ in range(K):
                p.execute(api.EXEC_NO_TIMING)
            p.sync()
            pipelined = (time.perf_counter() - t_host) / K * 1e3
            p.destroy()
            res[name].append((m[0], m[3], pipelined))
    for k in touched:
        os.environ.pop(k, None)
    print(f"# {n0}x{n1}x{n2} {prec} P={P}: per variant median over {reps} plans (min..max) of t0 ms, t3 ms, back-to-back ms per execute;"
          f" pass GB/s = 2 S N/P / t")
    for name, env in variants:
        m = np.array(res[name])
        t0, t3, pp = np.median(m[:, 0]), np.median(m[:, 1]), np.median(m[:, 2])
        bytes_pass = 2.0 * S * n0 * n1 * n2 / P
        print(f"{name:10s} t0 {t0:.4f} ({m[:, 0].min():.4f}..{m[:, 0].max():.4f})  t3 {t3:.4f} ({m[:, 1].min():.4f}..{m[:, 1].max():.4f})"
              f"  X pass {bytes_pass / t3 / 1e6:.0f} GB/s  t0 as two passes {2 * bytes_pass / t0 / 1e6:.0f} GB/s  back-to-back {pp:.4f}"
              f"  [{desc[name]}]", flush=True)
    if comm:
        comm.destroy()
    del a, b
    torch.cuda.empty_cache()
