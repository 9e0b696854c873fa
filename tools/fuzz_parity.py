"""Developer check (GPU box): random mid-size P > 1 transforms -- shapes whose slabs exceed the Infinity Cache, received planes a multiple of
128 KiB apart, uneven splits -- serial and overlapped pipeline, both precisions, against scipy.fft.fftn on the host, element by element;
the overlapped result must equal the serial one bit for bit wherever both plans run the same form of the YZ stage; every forward result goes
back through the backward plan of the same pipeline (round trip against the input).  P = 1 cases run the single-GPU plans twice.
(Where a destination block of the overlapped plan is narrower than 128 rows, Y axes of 1024 / 2048 points leave the DIF-split kernel for the
plain one: same tolerance, different last bits -- reported as "bits differ", not as a failure.)
usage: fuzz_parity.py [cases, default 24] [seed, default 1] [log2 of the largest problem, default 27]"""
import os
import sys
import threading
from pathlib import Path

import numpy as np
import scipy.fft as sf
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from distributedfft_amd import api  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
MAXLOG = int(sys.argv[3]) if len(sys.argv) > 3 else 27
dev = torch.device("cuda:0")
LEN = [64, 128, 192, 256, 384, 512, 640, 768, 1024, 2048]
TOL = {"f64": 1e-11, "f32": 5e-4}


def slab(n, P, g):
    blk = -(-n // P)
    return g * blk, max(0, min(n, (g + 1) * blk) - g * blk)


def run(N, P, prec, flags, x, direction=api.FORWARD, inputs=None):
    """forward: x is the natural [n0][n1][n2] array; backward: inputs[g] is device g's forward result (its whole buffer)"""
    n0, n1, n2 = N
    tdt = torch.complex128 if prec == "f64" else torch.complex64
    comm = api.Comm.local(P) if P > 1 else None
    plans, outs, keep = [], [], []
    for g in range(P):
        mc = api.get_max_data_count(n0, n1, n2, P, g == P - 1)
        x0, xs = slab(n0, P, g)
        a = torch.zeros(mc, dtype=tdt, device=dev)
        if inputs is None:
            a[:xs * n1 * n2] = torch.from_numpy(x[x0:x0 + xs].reshape(-1)).to(dev).to(tdt)
        else:
            a[:] = torch.from_numpy(inputs[g]).to(dev)
        b = torch.zeros(mc, dtype=tdt, device=dev)
        torch.cuda.synchronize()
        plans.append(api.Plan(n0, n1, n2, a, b, comm, g, P, direction, flags))
        outs.append(b)
        keep.append(a)
    desc = plans[0].describe()
    errs = []

    def work(g):
        try:
            for _ in range(2):  # twice: the counters of the one-launch stage run on from execute to execute
                plans[g].execute()
                plans[g].sync()
        except Exception as e:
            errs.append(e)
    th = [threading.Thread(target=work, args=(g,)) for g in range(P)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    res = [o.cpu().numpy() for o in outs]
    for p in plans:
        p.destroy()
    if comm:
        comm.destroy()
    return res, desc


bad = 0
done = 0
while done < cases:
    n0, n1, n2 = (int(rng.choice(LEN)) for _ in range(3))
    P = int(rng.choice([1, 2, 3, 4, 8]))
    prec = str(rng.choice(["f64", "f64", "f32"]))
    if n2 > 1024 or n0 * n1 * n2 > (1 << MAXLOG) or n0 * n1 * n2 < (1 << (MAXLOG - 5)):
        continue
    if slab(n0, P, P - 1)[1] < 1 or slab(n1, P, P - 1)[1] < 1:
        continue
    done += 1
    os.environ["DFFT_ROT"] = str(rng.choice(["", "1"]))  # automatic / forced on
    if not os.environ["DFFT_ROT"]:
        del os.environ["DFFT_ROT"]
    x = (rng.random((n0, n1, n2)) - 0.5 + 1j * (rng.random((n0, n1, n2)) - 0.5)).astype(np.complex128 if prec == "f64" else np.complex64)
    ref = sf.fftn(x.astype(np.complex128), workers=-1)
    scale = float(np.abs(ref).max())
    res = {}
    for name, flags in (("serial", api.PLAN_INPUT_FROM_IN), ("overlap", api.PLAN_INPUT_FROM_IN | (api.PLAN_OVERLAP if P > 1 else 0))):
        out, desc = run((n0, n1, n2), P, prec, flags, x)
        worst = 0.0
        for d in range(P):
            y0, ys = slab(n1, P, d)
            exp = np.transpose(ref[:, y0:y0 + ys, :], (1, 2, 0))
            got = out[d][:ys * n2 * n0].reshape(ys, n2, n0)
            worst = max(worst, float(np.abs(got - exp).max()) / scale)
        # round trip through the backward plan of the same pipeline: backward(forward(x)) / N == x
        back, _ = run((n0, n1, n2), P, prec, flags, None, api.BACKWARD, out)
        rt = 0.0
        for g in range(P):
            x0, xs = slab(n0, P, g)
            rt = max(rt, float(np.abs(back[g][:xs * n1 * n2].reshape(xs, n1, n2) / (n0 * n1 * n2) - x[x0:x0 + xs]).max()))
        worst = max(worst, rt / float(np.abs(x).max()))
        res[name] = (out, desc, worst)
    same_form = res["serial"][1].split("yz_stage=")[1].split()[0] == res["overlap"][1].split("yz_stage=")[1].split()[0]
    bits = all(np.array_equal(a, b) for a, b in zip(res["serial"][0], res["overlap"][0]))
    ok = res["serial"][2] < TOL[prec] and res["overlap"][2] < TOL[prec]
    bad += 0 if ok else 1
    print(f"{('ok  ' if bits or not same_form else 'ok (bits differ)') if ok else 'FAIL'} {n0}x{n1}x{n2} {prec} P={P} rot={os.environ.get('DFFT_ROT', 'auto')}  err serial {res['serial'][2]:.2e} overlap {res['overlap'][2]:.2e}"
          f"  overlap==serial bits: {bits} (same form: {same_form})  [{res['overlap'][1].split('handover=')[0].strip()} | "
          f"{' '.join(t for t in res['overlap'][1].split() if t.startswith(('rotated', 'overlap_parts', 'ysub', 'parts_in')))}]", flush=True)
    del x, ref, res
    torch.cuda.empty_cache()
print(f"{done} cases, {bad} failed")
sys.exit(1 if bad else 0)
