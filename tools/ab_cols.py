"""Developer A/B (GPU): column-kernel rates of the 20- / 24-point plans and the 3D shapes built on them, for the library
selected by DFFT_LIB (default: the in-tree build).   python tools/ab_cols.py [label]"""
import math, os, sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("DFFT_TUNE", "0")
from distributedfft_amd import _lib as L, api

label = sys.argv[1] if len(sys.argv) > 1 else "base"
dev = torch.device("cuda:0")
lib = L.load()
s = torch.cuda.current_stream().cuda_stream


def tl(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for dtype, code, S in ((torch.complex128, 0, 16), (torch.complex64, 1, 8)):
    for n in (96, 192, 320, 384, 400, 640, 768, 512):
        width = 512
        b2 = max(1, (1 << 26) // (n * width))
        x = torch.rand(b2 * n * width, dtype=torch.float64, device=dev).to(dtype)
        y = torch.empty_like(x)
        ms = tl(lambda: lib.dfft_fft1d_cols(x.data_ptr(), y.data_ptr(), n, width, b2, code, 1, s))
        print(f"{label} cols n={n:5d} {'f64' if code == 0 else 'f32'}: {ms:.4f} ms  {2 * S * b2 * n * width / ms / 1e6:6.0f} GB/s", flush=True)
        del x, y
for N, dtype in (((384, 384, 384), torch.complex128), ((768, 768, 768), torch.complex128), ((1024, 768, 512), torch.complex128),
                 ((640, 640, 640), torch.complex128), ((768, 768, 768), torch.complex64), ((1024, 768, 512), torch.complex64)):
    n = N[0] * N[1] * N[2]
    rdt = torch.float64 if dtype == torch.complex128 else torch.float32
    a = torch.complex(torch.rand(n, device=dev, dtype=rdt), torch.rand(n, device=dev, dtype=rdt))
    b = torch.zeros_like(a)
    p = api.Plan(*N, a, b, None, 0, 1, api.FORWARD, api.PLAN_INPUT_FROM_IN)
    for _ in range(3):
        p.execute(api.EXEC_NO_TIMING)
    ts = []
    for _ in range(9):
        p.execute(api.EXEC_ASYNC)
        ts.append(p.stage_times())
    ts = np.median(np.array(ts), axis=0) * 1e3
    print(f"{label} 3d {'x'.join(map(str, N)):>13} {'f64' if dtype == torch.complex128 else 'f32'}: t0 {ts[0]:.4f}  t3 {ts[3]:.4f}  total {ts.sum():.4f} ms", flush=True)
    p.destroy()
    del a, b
    torch.cuda.empty_cache()
