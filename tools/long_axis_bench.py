"""Developer timing (GPU box): the long-axis kernels of BASELINE config 5 -- 1024 / 2048-point rows and columns, both
precisions, and the single-GPU 3D pipeline on shapes with a 2048-point axis.  CSV to stdout."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import sweep_bench as sb  # noqa: E402

if __name__ == "__main__":
    sb.LENGTHS = [512, 1000, 1024, 1280, 1536, 2048]
    sb.sweep_1d()
    import math
    import numpy as np
    import torch
    from distributedfft_amd import api
    print("shape,dtype,t0_ms,t3_ms,total_ms,GFlops,t0_GBps,t3_GBps")
    for dtype, S in ((torch.complex128, 16), (torch.complex64, 8)):
        for N in [(2048, 1024, 512), (512, 2048, 512), (1024, 768, 512), (2048, 256, 1024), (256, 2048, 1024)]:
            n = N[0] * N[1] * N[2]
            a = torch.rand(n, dtype=torch.float64 if S == 16 else torch.float32, device=sb.DEV).to(dtype)
            b = torch.zeros_like(a)
            torch.cuda.synchronize()
            plan = api.Plan(*N, a, b, None, 0, 1, api.FORWARD, api.PLAN_INPUT_FROM_IN)
            for _ in range(2):
                plan.execute()
            plan.sync()
            ts = []
            for _ in range(7):
                plan.execute()
                ts.append(plan.stage_times())
            ts = np.array(ts)
            med = np.median(ts, axis=0)
            tot = float(np.median(ts.sum(axis=1)))
            print(f"{N[0]}x{N[1]}x{N[2]},{'f64' if S == 16 else 'f32'},{med[0] * 1e3:.4f},{med[3] * 1e3:.4f},{tot * 1e3:.4f},"
                  f"{5.0 * n * math.log2(n) * 1e-9 / tot:.0f},{4 * S * n / med[0] / 1e9:.0f},{2 * S * n / med[3] / 1e9:.0f}", flush=True)
            plan.destroy()
            del a, b
            torch.cuda.empty_cache()
