"""Developer sweep (GPU box): effective HBM rate of every plan length (rows / columns, fp64 / fp32) and of the single-GPU 3D
pipeline over a range of shapes.  Writes CSV to stdout.  Not the graded bench."""
import math
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from distributedfft_amd import _lib as L  # noqa: E402
from distributedfft_amd import api  # noqa: E402

LENGTHS = [2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 16, 24, 25, 32, 48, 49, 64, 96, 100, 125, 128, 192, 256, 343, 384, 512, 768,
           1024, 2048]
DEV = torch.device("cuda:0")


def time_launch(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def sweep_1d(total_elems=1 << 26):
    lib = L.load()
    s = torch.cuda.current_stream().cuda_stream
    print("kind,n,dtype,batch,width,ms,GBps")
    for dtype, code, S in ((torch.complex128, 0, 16), (torch.complex64, 1, 8)):
        for n in LENGTHS:
            batch = total_elems // n
            x = torch.rand(batch * n, dtype=torch.float64, device=DEV).to(dtype)
            y = torch.empty_like(x)
            ms = time_launch(lambda: lib.dfft_fft1d_rows(x.data_ptr(), y.data_ptr(), n, batch, code, 1, s))
            print(f"rows,{n},{'f64' if code == 0 else 'f32'},{batch},,{ms:.4f},{2 * S * batch * n / ms / 1e6:.0f}", flush=True)
            width = 512
            b2 = max(1, total_elems // (n * width))
            ms = time_launch(lambda: lib.dfft_fft1d_cols(x.data_ptr(), y.data_ptr(), n, width, b2, code, 1, s))
            print(f"cols,{n},{'f64' if code == 0 else 'f32'},{b2},{width},{ms:.4f},{2 * S * b2 * n * width / ms / 1e6:.0f}", flush=True)
            del x, y


def sweep_3d():
    shapes = [(64, 64, 64), (96, 96, 96), (100, 100, 100), (128, 128, 128), (192, 192, 192), (256, 256, 256), (343, 343, 343),
              (384, 384, 384), (512, 512, 512), (768, 768, 768), (1024, 1024, 1024), (1024, 768, 512), (2048, 512, 512),
              (512, 2048, 512), (512, 512, 2048), (2048, 1024, 512)]
    # total_ms: sum of the stage times of single executes (HIP events between the stages); pipelined_ms: 50 back-to-back
    # executes without stage events (DFFT_EXEC_NO_TIMING: graph replay) bracketed by one pair of events, per execute
    print("shape,dtype,t0_ms,t3_ms,total_ms,GFlops,t0_GBps,t3_GBps,pipelined_ms,pipelined_GFlops")
    for dtype, S in ((torch.complex128, 16), (torch.complex64, 8)):
        for N in shapes:
            n = N[0] * N[1] * N[2]
            a = torch.rand(n, dtype=torch.float64 if S == 16 else torch.float32, device=DEV).to(dtype)
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
            for _ in range(3):
                plan.execute(api.EXEC_NO_TIMING)
            plan.sync()
            ps = torch.cuda.ExternalStream(plan.stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 50 if n <= 512 ** 3 else 10
            e0.record(ps)
            for _ in range(reps):
                plan.execute(api.EXEC_NO_TIMING)
            e1.record(ps)
            plan.sync()
            pip = e0.elapsed_time(e1) / reps
            print(f"{N[0]}x{N[1]}x{N[2]},{'f64' if S == 16 else 'f32'},{med[0] * 1e3:.4f},{med[3] * 1e3:.4f},{tot * 1e3:.4f},"
                  f"{5.0 * n * math.log2(n) * 1e-9 / tot:.0f},{4 * S * n / med[0] / 1e9:.0f},{2 * S * n / med[3] / 1e9:.0f},"
                  f"{pip:.4f},{5.0 * n * math.log2(n) * 1e-6 / pip:.0f}", flush=True)
            plan.destroy()
            del a, b
            torch.cuda.empty_cache()


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("1d", "all"):
        sweep_1d()
    if which in ("3d", "all"):
        sweep_3d()
