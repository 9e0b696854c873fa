"""Developer check (GPU box): single-GPU forward -> backward round trip of shapes whose backward plan runs the chunk loop of the inverse YZ
stage on a hand-over buffer (fp32, or fp64 with a long Y / Z axis), against the input, and the backward result against scipy's inverse of
the forward result.   usage: roundtrip_check.py [n0xn1xn2:prec ...]"""
import sys
from pathlib import Path

import numpy as np
import scipy.fft as sf
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from distributedfft_amd import api  # noqa: E402

dev = torch.device("cuda:0")
specs = sys.argv[1:] or ["512x512x512:f32", "1024x512x512:f32", "512x1024x512:f64", "1024x768x512:f32", "256x2048x512:f32"]
bad = 0
for spec in specs:
    shape, prec = spec.split(":")
    n0, n1, n2 = (int(v) for v in shape.split("x"))
    N = n0 * n1 * n2
    cdt, tdt = (np.complex128, torch.complex128) if prec == "f64" else (np.complex64, torch.complex64)
    rng = np.random.default_rng(5)
    x = (rng.random((n0, n1, n2), dtype=np.float32) - 0.5 + 1j * (rng.random((n0, n1, n2), dtype=np.float32) - 0.5)).astype(cdt)
    a = torch.from_numpy(x.reshape(-1)).to(dev)
    b = torch.zeros_like(a)
    c = torch.zeros_like(a)
    torch.cuda.synchronize()
    f = api.Plan(n0, n1, n2, a, b, None, 0, 1, api.FORWARD, api.PLAN_INPUT_FROM_IN)
    f.execute(); f.sync()
    g = api.Plan(n0, n1, n2, b, c, None, 0, 1, api.BACKWARD, api.PLAN_INPUT_FROM_IN)
    g.execute(); g.sync()
    desc = g.describe()
    back = c.cpu().numpy().reshape(n0, n1, n2)
    rt = float(np.abs(back / N - x).max() / np.abs(x).max())
    # the backward transform by itself: input [y][z][kx] = the forward result, expected = unnormalised inverse of it in [x][y][z]
    fw = b.cpu().numpy().reshape(n1, n2, n0)
    exp = sf.ifftn(np.transpose(fw, (2, 0, 1)).astype(np.complex128), workers=-1) * N
    alone = float(np.abs(back - exp).max() / np.abs(exp).max())
    tol = 1e-11 if prec == "f64" else 5e-4
    ok = rt < tol and alone < tol
    bad += 0 if ok else 1
    print(f"{'ok  ' if ok else 'FAIL'} {spec}  round trip {rt:.2e}  backward alone {alone:.2e}  [{desc.split('handover=')[0].strip()} handover={desc.split('handover=')[1].split()[0]}]", flush=True)
    f.destroy(); g.destroy()
    del a, b, c, x, back, fw, exp
    torch.cuda.empty_cache()
print(f"{len(specs)} shapes, {bad} failed")
sys.exit(1 if bad else 0)
