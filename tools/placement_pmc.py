"""Developer probe (GPU box): what distinguishes a "fast" from a "slow" hand-over buffer of the 512^3 fp64 X pass?

Creates NPLANS plans (kept alive, so every one gets different physical memory for its hand-over buffer W), runs each the same
number of executes and prints per plan: median t0 / t3, the time of an in-place streaming pass over W (dfft_scale with s = 1: a
contiguous 2 GiB read + 2 GiB write on exactly those pages) and of the same pass over the caller's output buffer.  Run it
plainly for the timings and under `rocprofv3 --kernel-trace --pmc ...` for per-dispatch counters: the X-pass dispatches appear
in plan order, EXECS per plan.   DFFT_W_ALLOC selects how W is allocated (dfft_alloc.cpp).
usage: placement_pmc.py [nplans] [execs] [modes,comma,separated]"""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("DFFT_TUNE", "0")
from distributedfft_amd import api, _lib as L  # noqa: E402

nplans = int(sys.argv[1]) if len(sys.argv) > 1 else 6
execs = int(sys.argv[2]) if len(sys.argv) > 2 else 16
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else [os.environ.get("DFFT_W_ALLOC", "malloc")]
dev = torch.device("cuda:0")
n = 512
a = (torch.rand(n ** 3, device=dev, dtype=torch.float64) * 2 - 1).to(torch.complex128)
b = torch.zeros_like(a)
lib = L.load()


def stream_pass(ptr, count, reps=5):
    """in-place x *= 1.0 over `count` fp64 complex elements: median ms"""
    s = torch.cuda.current_stream().cuda_stream
    ts = []
    for _ in range(reps + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(lib.dfft_scale(ptr, count, 0, 1.0, s), "dfft_scale")
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[2:]))


for mode in modes:
    os.environ["DFFT_W_ALLOC"] = mode
    plans = []
    for i in range(nplans):
        p = api.Plan(n, n, n, a, b, None, 0, 1, api.FORWARD, api.PLAN_INPUT_FROM_IN)
        plans.append(p)
        for _ in range(execs - 6):
            p.execute(api.EXEC_NO_TIMING)
        ts = []
        for _ in range(6):
            p.execute(api.EXEC_ASYNC)
            ts.append(p.stage_times())
        ts = np.median(np.array(ts), axis=0) * 1e3
        nb = C.c_longlong(0)
        w = lib.dfft_plan_workbuf(p.handle, C.byref(nb))
        sw = stream_pass(w, n ** 3) if w else float("nan")
        so = stream_pass(b.data_ptr(), n ** 3)
        print("mode %-12s plan %d: t0 %.4f  t3 %.4f ms | W@%x  stream(W) %.4f  stream(out) %.4f ms" % (mode, i, ts[0], ts[3], w or 0, sw, so), flush=True)
    for p in plans:
        p.destroy()
    torch.cuda.synchronize()
