import sys, math, numpy as np, torch
sys.path.insert(0, ".")
from distributedfft_amd import api
for dtype, S in ((torch.complex128, 16), (torch.complex64, 8)):
    for N in [(2048, 1024, 512), (2048, 256, 1024)]:
        n = N[0] * N[1] * N[2]
        a = torch.rand(n, dtype=torch.float64 if S == 16 else torch.float32, device="cuda").to(dtype)
        b = torch.zeros_like(a)
        plan = api.Plan(*N, a, b, None, 0, 1, api.FORWARD, api.PLAN_INPUT_FROM_IN)
        for _ in range(2): plan.execute()
        plan.sync()
        ts = []
        for _ in range(9):
            plan.execute(); ts.append(plan.stage_times())
        med = np.median(np.array(ts), axis=0)
        print(f"{N} {'f64' if S==16 else 'f32'} t3 {med[3]*1e3:.3f} ms {2*S*n/med[3]/1e9:.0f} GB/s", flush=True)
        plan.destroy(); del a, b; torch.cuda.empty_cache()
