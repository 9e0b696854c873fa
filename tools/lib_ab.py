"""Developer A/B (GPU box): one LIBRARY BUILD per process (DFFT_LIB=<path>, e.g. a tools/build_variant.py build), a list of plans.
Prints per plan the sha256 of the forward result (bit-identity between builds: compare the digests of two runs), t0 / t3 (median of 9
timed executes) and the un-timed back-to-back rate.  Run the builds alternately a few times: a plan's buffers land in different
physical regions from one process to the next (5-8 % of an X pass, profiles/r03/README.md section 1); single-GPU plans are placed with
dfft_plan_tune first.

usage: lib_ab.py SPEC [SPEC ...]     SPEC = n0xn1xn2:prec:P[:ENV=val+ENV=val]      (P > 1: rank 0's local work, exchange switched off)"""
import hashlib
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
lib = os.environ.get("DFFT_LIB", "default")
for spec in specs:
    parts = spec.split(":")
    n0, n1, n2 = (int(v) for v in parts[0].split("x"))
    prec, P = parts[1], int(parts[2])
    env = dict(e.split("=", 1) for e in parts[3].split("+") if e) if len(parts) > 3 else {}
    os.environ.update(env)
    cdt = torch.complex128 if prec == "fp64" else torch.complex64
    S = 16 if prec == "fp64" else 8
    mc = api.get_max_data_count(n0, n1, n2, P, False)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    a = (torch.rand(mc, generator=g, device=dev, dtype=torch.float32) - 0.5).to(cdt)
    b = torch.zeros_like(a)
    comm = api.Comm.local(P) if P > 1 else None
    direction = api.BACKWARD if os.environ.get("DFFT_AB_DIR") == "-1" else api.FORWARD  # (backward: the four stage times are printed in execution order)
    extra = api.PLAN_UNFUSED if os.environ.get("DFFT_AB_UNFUSED") == "1" else 0  # (the reference's stage structure: separate pack / transpose launches)
    p = api.Plan(n0, n1, n2, a, b, comm, 0, P, direction, api.PLAN_INPUT_FROM_IN | extra)
    if P == 1:
        p.tune()
    for _ in range(6):
        p.execute(api.EXEC_NO_TIMING)
    p.sync()
    digest = hashlib.sha256(b.cpu().numpy().tobytes()).hexdigest()[:16]
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
    bytes_pass = 2.0 * S * n0 * n1 * n2 / P
    print(f"{Path(lib).name:28s} {spec:44s} sha {digest}  t0 {m[0]:.4f}  t3 {m[3]:.4f}  X pass {bytes_pass / m[3] / 1e6:.0f} GB/s"
          f"  t0 as two passes {2 * bytes_pass / m[0] / 1e6:.0f} GB/s  back-to-back {pipelined:.4f}  [{p.describe()}]"
          + (f"  backward stages {m[0]:.4f} {m[1]:.4f} {m[2]:.4f} {m[3]:.4f}" if direction == api.BACKWARD else ""), flush=True)
    p.destroy()
    if comm:
        comm.destroy()
    for k in env:
        os.environ.pop(k, None)
    del a, b
    torch.cuda.empty_cache()
