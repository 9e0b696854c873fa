import os, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from distributedfft_amd import api
from oracle import slab_oracle as so
dev = torch.device("cuda:0")
os.environ["DFFT_PAD"] = "1"
for N, chunk in (((16, 256, 256), 5), ((12, 512, 256), 4), ((70, 512, 512), 64), ((512, 256, 256), 0)):
    if chunk: os.environ["DFFT_CHUNK_PLANES"] = str(chunk)
    else: os.environ.pop("DFFT_CHUNK_PLANES", None)
    n = N[0] * N[1] * N[2]
    x = so.random_input(N, seed=N[0] + 3)
    a = torch.from_numpy(x.reshape(-1)).to(dev)
    ref = np.fft.fftn(x).transpose(1, 2, 0).reshape(-1) if n <= 1 << 25 else None
    outs = {}
    for mode in ("1", "lazy", "lazy2"):
        os.environ["DFFT_T0_ONE_LAUNCH"] = "1"
        os.environ["DFFT_ZY_LAZY"] = "0" if mode == "1" else "1"
        b, c = torch.zeros_like(a), torch.zeros_like(a)
        p = api.Plan(*N, a, b, None, 0, 1, api.FORWARD, api.PLAN_INPUT_FROM_IN)
        q = api.Plan(*N, b, c, None, 0, 1, api.BACKWARD, api.PLAN_INPUT_FROM_IN)
        for _ in range(3): p.execute(api.EXEC_NO_TIMING)
        p.sync()
        for _ in range(2): q.execute(api.EXEC_NO_TIMING)
        q.sync()
        outs[mode] = (b.clone(), c.clone())
        p.destroy(); q.destroy()
    for k, nm in ((0, "fwd"), (1, "bwd")):
        d = (outs["1"][k] - outs["lazy"][k]).abs()
        d2 = (outs["lazy"][k] - outs["lazy2"][k]).abs()
        sc = outs["1"][k].abs().max().item()
        nz = int((d > 0).sum().item())
        msg = f"{N} {nm}: base-vs-lazy max|diff|/max {d.max().item() / sc:.3e} differing {nz}/{n}  lazy-vs-lazy2 max {d2.max().item() / sc:.3e} differing {int((d2 > 0).sum().item())}"
        if nz:
            idx = torch.nonzero(d > 0).flatten()[:6].tolist()
            msg += f" first idx {idx}"
        if ref is not None and k == 0:
            msg += f"  base-vs-numpy {np.abs(outs['1'][0].cpu().numpy() - ref).max() / np.abs(ref).max():.3e} lazy-vs-numpy {np.abs(outs['lazy'][0].cpu().numpy() - ref).max() / np.abs(ref).max():.3e}"
        print(msg, flush=True)
