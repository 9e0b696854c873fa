"""Developer probe: is the bimodal X-pass time (0.71 vs 0.77 ms at 512^3 fp64) tied to where a buffer was allocated?
Same process, same input: (a) re-create the plan (new bufferDev1 / padded buffer) with the same in/out tensors, (b) re-allocate
the output tensor and re-create the plan, (c) allocate spacer blocks of varying size before creating the plan."""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from distributedfft_amd import api

dev = torch.device("cuda:0")
n = 512
a = (torch.rand(n ** 3, device=dev, dtype=torch.float64) * 2 - 1).to(torch.complex128)
b = torch.zeros_like(a)


def t3_of(plan):
    for _ in range(3):
        plan.execute(api.EXEC_NO_TIMING)
    ts = []
    for _ in range(5):
        plan.execute(api.EXEC_ASYNC)
        ts.append(plan.stage_times())
    ts = np.median(np.array(ts), axis=0)
    return ts[0] * 1e3, ts[3] * 1e3


print("(a) same in/out, plan re-created")
for i in range(6):
    p = api.Plan(n, n, n, a, b, None, 0, 1, api.FORWARD, api.PLAN_INPUT_FROM_IN)
    print("  plan %d: t0 %.4f  t3 %.4f ms   out@%x" % ((i,) + t3_of(p) + (b.data_ptr(),)))
    p.destroy()
print("(a2) plans kept alive (every new plan gets new buffers)")
plans = []
for i in range(8):
    p = api.Plan(n, n, n, a, b, None, 0, 1, api.FORWARD, api.PLAN_INPUT_FROM_IN)
    plans.append(p)
    print("  plan %d: t0 %.4f  t3 %.4f ms" % ((i,) + t3_of(p)))
for p in plans:
    p.destroy()
print("(b) output tensor re-allocated")
keep = []
for i in range(6):
    b2 = torch.zeros_like(a)
    keep.append(b2)          # keep the old ones alive so that every b2 is a new address
    p = api.Plan(n, n, n, a, b2, None, 0, 1, api.FORWARD, api.PLAN_INPUT_FROM_IN)
    print("  out  %d: t0 %.4f  t3 %.4f ms   out@%x" % ((i,) + t3_of(p) + (b2.data_ptr(),)))
    p.destroy()
del keep
torch.cuda.empty_cache()
print("(c) spacer of k MiB allocated (and kept) before the plan")
for k in (1, 3, 7, 64, 65, 129, 1023):
    sp = torch.empty(k << 20, dtype=torch.uint8, device=dev)
    p = api.Plan(n, n, n, a, b, None, 0, 1, api.FORWARD, api.PLAN_INPUT_FROM_IN)
    print("  spacer %4d MiB: t0 %.4f  t3 %.4f ms" % ((k,) + t3_of(p)))
    p.destroy()
    del sp
