"""-m gpu: the BASELINE.json configurations at FULL size, checked through size-independent properties (the oracle cannot
finish these in seconds): separable plane waves -> isolated spikes of height N at the right [yy][z][kx] position of the
right device, impulse -> plane wave, Parseval checksum, and forward/backward round trip.  P > 1 runs as P virtual devices
on the one GPU (in-process exchange), i.e. the decomposition, packing and offsets of the real multi-GPU layout.

  C2  256^3 fp64  P=1        512^3 fp64  P=1 and P=4        C4  1024x768x512 fp64  P=8 (radix-3 Y axis, non-cubic)
  C5  2048x2048x1024 fp32  P=8 (2^32 elements, 64-bit indexing, 2048-point tiles)
"""
import math
import threading

import pytest

pytestmark = pytest.mark.gpu

CONFIGS = [
    pytest.param((256, 256, 256), 1, "f64", False, id="C2-256^3-fp64-P1"),
    pytest.param((512, 512, 512), 1, "f64", False, id="512^3-fp64-P1"),
    pytest.param((512, 512, 512), 4, "f64", False, id="C3-512^3-fp64-P4"),
    pytest.param((1024, 768, 512), 8, "f64", False, id="C4-1024x768x512-fp64-P8"),
    pytest.param((2048, 2048, 1024), 8, "f32", False, id="C5-2048x2048x1024-fp32-P8"),
    # the overlapped pipeline (X-plane parts, Y sub-blocks) at full size -- what bench.py times for P > 1
    pytest.param((512, 512, 512), 4, "f64", True, id="C3-512^3-fp64-P4-overlap"),
    pytest.param((512, 512, 512), 8, "f64", True, id="512^3-fp64-P8-overlap"),
    pytest.param((2048, 2048, 1024), 8, "f32", True, id="C5-2048x2048x1024-fp32-P8-overlap"),
    # config 4 overlapped: at P = 8 the X-plane parts are sized like cache chunks (3 x 40 + 8 planes, two launches per part), at P = 4 ONE
    # launch of the YZ stage serves all parts on 96-row Y sub-blocks
    pytest.param((1024, 768, 512), 8, "f64", True, id="C4-1024x768x512-fp64-P8-overlap"),
    pytest.param((1024, 768, 512), 4, "f64", True, id="C4-1024x768x512-fp64-P4-overlap"),
]
TOL = {"f64": 1e-11, "f32": 5e-4}


def _slab(n, P, g):
    blk = -(-n // P)
    return g * blk, (blk if g < P - 1 else n - (P - 1) * blk)


def _run(plans):
    errs = []

    def work(p):
        try:
            p.execute()
            p.sync()
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=work, args=(p,)) for p in plans]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs


@pytest.mark.parametrize("N,P,prec,overlap", CONFIGS)
def test_fullsize_properties(gpu, N, P, prec, overlap):
    import torch
    from distributedfft_amd import api
    n0, n1, n2 = N
    NT = n0 * n1 * n2
    cdt = torch.complex128 if prec == "f64" else torch.complex64
    # HBM really needed at the peak: per device the caller's in + out, the plan's bufferDev1, and either the overlapped
    # exchange's receive buffer (forward) or the round trip's result (backward) = 4 slabs; plus the slab-sized temporaries
    # of the input synthesis / residual arithmetic below (at most 6 alive at once, one device at a time)
    torch.cuda.empty_cache()  # blocks cached by earlier tests do not count as free otherwise
    free, _ = torch.cuda.mem_get_info()
    S = 16 if prec == "f64" else 8
    slabs = [api.get_max_data_count(n0, n1, n2, P, g == P - 1) * S for g in range(P)]
    need = 5 * sum(slabs) + 6 * max(slabs) + (1 << 30)  # + the plan's padded work buffer
    if need > free:
        pytest.skip(f"needs {need / 2**30:.0f} GiB of HBM, {free / 2**30:.0f} free")

    # three separable plane waves with amplitudes a_m at (kx, ky, kz) chosen to land on different devices / corners
    waves = [(1.0, (1, 0, 0)), (0.5, (n0 - 1, n1 // 2 + 1, 3)), (0.25, (n0 // 3, n1 - 1, n2 - 1))]

    def axis(n, k, lo, cnt):
        t = torch.arange(lo, lo + cnt, device=gpu, dtype=torch.float64)
        ph = 2 * math.pi * ((k * t) % n) / n
        return torch.complex(torch.cos(ph), torch.sin(ph)).to(cdt)

    comm = api.Comm.local(P) if P > 1 else None
    ins, outs, plans = [], [], []
    energy_in = 0.0
    for g in range(P):
        x0, xs = _slab(n0, P, g)
        mc = api.get_max_data_count(n0, n1, n2, P, g == P - 1)
        a = torch.zeros(mc, dtype=cdt, device=gpu)
        v = a[:xs * n1 * n2].view(xs, n1, n2)
        for amp, (kx, ky, kz) in waves:
            v += amp * axis(n0, kx, x0, xs)[:, None, None] * axis(n1, ky, 0, n1)[None, :, None] * axis(n2, kz, 0, n2)[None, None, :]
        if g == 0:
            v[0, 0, 0] += 1.0  # plus an impulse at the origin -> +1 on every output bin
        energy_in += float((v.real.double() ** 2 + v.imag.double() ** 2).sum().item())
        b = torch.zeros(mc, dtype=cdt, device=gpu)
        ins.append(a)
        outs.append(b)
        plans.append(api.Plan(n0, n1, n2, a, b, comm, g, P, api.FORWARD,
                              api.PLAN_INPUT_FROM_IN | (api.PLAN_OVERLAP if overlap else 0)))
    if overlap and N == (1024, 768, 512):   # the plan really is what the case is listed for (dfft_plan_describe)
        desc = plans[0].describe()
        want = "overlap_parts=40 " if P == 8 else "parts_in_one_launch=1 "
        assert want in desc and ("yz_stage=one-launch-lazy" in desc) == (P == 4), desc
    _run(plans)

    # expected: 1 everywhere (impulse) + a_m * N at (kx, ky, kz); layout out_d[yy][z][kx]
    tol = TOL[prec] * NT
    energy_out = 0.0
    for d in range(P):
        y0, ys = _slab(n1, P, d)
        o = outs[d][:ys * n2 * n0].view(ys, n2, n0)
        energy_out += float((o.real.double() ** 2 + o.imag.double() ** 2).sum().item())
        o = o - 1.0
        for amp, (kx, ky, kz) in waves:
            if y0 <= ky < y0 + ys:
                got = o[ky - y0, kz, kx]
                assert abs(complex(got.item()) - amp * NT) < tol, (d, kx, ky, kz, got.item())
                o[ky - y0, kz, kx] -= amp * NT
        resid = float(o.abs().max().item())
        assert resid < tol, f"device {d}: residual {resid:.3e} (tol {tol:.3e})"
        del o
    # Parseval: sum |X|^2 = N sum |x|^2 (a checksum over every output element)
    assert abs(energy_out / (NT * energy_in) - 1.0) < (1e-10 if prec == "f64" else 1e-4)

    # round trip: backward(forward(x)) / N == x
    for p in plans:
        p.destroy()
    backs, bplans = [], []
    for g in range(P):
        c = torch.zeros_like(ins[g])
        backs.append(c)
        bplans.append(api.Plan(n0, n1, n2, outs[g], c, comm, g, P, api.BACKWARD, api.PLAN_INPUT_FROM_IN))
    _run(bplans)
    for g in range(P):
        x0, xs = _slab(n0, P, g)
        cnt = xs * n1 * n2
        err = float((backs[g][:cnt] / NT - ins[g][:cnt]).abs().max().item())
        assert err < (1e-11 if prec == "f64" else 2e-4), f"round trip device {g}: {err:.3e}"
    for p in bplans:
        p.destroy()
    if comm:
        comm.destroy()


# ---- element-wise parity at the graded size -------------------------------------------------------------------------------
# Every output element of the full-size transform against an independent host FFT (pocketfft through scipy, all host cores;
# numpy.fft as the fall-back) of the same input: the property checks above would miss a small error confined to a few bins.
def _host_fftn(x):
    try:
        import scipy.fft as sf
        return sf.fftn(x, workers=-1)
    except Exception:  # pragma: no cover
        import numpy as np
        return np.fft.fftn(x)


ELEMENTWISE = [
    pytest.param((512, 512, 512), 1, False, id="512^3-fp64-P1-elementwise"),             # the benchmarked configuration
    pytest.param((512, 512, 512), 4, True, id="C3-512^3-fp64-P4-overlap-elementwise"),   # what bench.py times at 4 GPUs
    pytest.param((1024, 768, 512), 8, False, id="C4-1024x768x512-fp64-P8-elementwise"),  # radix-3 axis, non-cubic slabs
]


@pytest.mark.parametrize("N,P,overlap", ELEMENTWISE)
def test_fullsize_elementwise_vs_host_fft(gpu, N, P, overlap, monkeypatch):
    import numpy as np
    import torch
    from distributedfft_amd import api
    # P > 1: with the opt-in placement of the receive buffer (dfft_plan.cpp, place_recv_buffer) -- the slabs of these shapes are the
    # ones it applies to (larger than the Infinity Cache); a few candidates are enough to exercise the swap of the plan's buffer
    monkeypatch.setenv("DFFT_TUNE_RECV", "1")
    monkeypatch.setenv("DFFT_TUNE_TRIES", "6")
    n0, n1, n2 = N
    NT = n0 * n1 * n2
    torch.cuda.empty_cache()
    comm = api.Comm.local(P) if P > 1 else None
    gen = torch.Generator(device=gpu)
    gen.manual_seed(20260921)
    ins, outs, plans = [], [], []
    host = np.empty((n0, n1, n2), dtype=np.complex128)
    for g in range(P):
        x0, xs = _slab(n0, P, g)
        mc = api.get_max_data_count(n0, n1, n2, P, g == P - 1)
        a = torch.zeros(mc, dtype=torch.complex128, device=gpu)
        cnt = xs * n1 * n2
        a[:cnt] = torch.complex(torch.rand(cnt, generator=gen, device=gpu, dtype=torch.float64) * 2 - 1,
                                torch.rand(cnt, generator=gen, device=gpu, dtype=torch.float64) * 2 - 1)
        host[x0:x0 + xs] = a[:cnt].view(xs, n1, n2).cpu().numpy()
        b = torch.zeros(mc, dtype=torch.complex128, device=gpu)
        ins.append(a)
        outs.append(b)
        plans.append(api.Plan(n0, n1, n2, a, b, comm, g, P, api.FORWARD,
                              api.PLAN_INPUT_FROM_IN | (api.PLAN_OVERLAP if overlap else 0)))
    _run(plans)
    ref = _host_fftn(host)            # [kx][ky][kz]
    del host
    scale = float(np.abs(ref).max())
    worst = 0.0
    for d in range(P):
        y0, ys = _slab(n1, P, d)
        got = outs[d][:ys * n2 * n0].view(ys, n2, n0).cpu().numpy()       # [yy][z][kx]
        want = ref[:, y0:y0 + ys, :].transpose(1, 2, 0)                   # the same view of the host result
        worst = max(worst, float(np.abs(got - want).max()))
        del got
    for p in plans:
        p.destroy()
    if comm:
        comm.destroy()
    assert worst / scale <= 1e-11, f"{NT} elements: max |diff| / max |ref| = {worst / scale:.3e}"


# ---- element-wise evidence for config 5 (2^32 points, fp32): the defining sum, evaluated in fp64 on the GPU -----------------
# A host FFT of 2^32 points is out of reach, so 4096 output bins -- a seeded 16 x 16 x 16 product grid whose ky values are spread
# over all 8 devices' Y-slabs -- are computed from the DFT's definition over the whole uniform-random input (three fp64
# contractions, ~6e11 flops) and compared bin by bin.  heFFTe's own bar for this problem is a random world compared bin by bin at
# 5e-4 for fp32 (heffte/heffteBenchmark/test/test_fft3d.h:20-28,101-108, test_common.h:136-151: max |a - b| / max |b|).
@pytest.mark.parametrize("overlap", [False, True], ids=["C5-direct-sum", "C5-direct-sum-overlap"])
def test_c5_bins_against_the_defining_sum(gpu, overlap):
    import torch
    from distributedfft_amd import api
    n0, n1, n2 = 2048, 2048, 1024
    P, KB = 8, 16
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    slab = api.get_max_data_count(n0, n1, n2, P, False) * 8
    need = (4 if overlap else 3) * P * slab + 4 * slab + (4 << 30)
    if need > free:
        pytest.skip(f"needs {need / 2**30:.0f} GiB of HBM, {free / 2**30:.0f} free")
    gen = torch.Generator(device=gpu)
    gen.manual_seed(4242)
    cpu = torch.Generator()
    cpu.manual_seed(4242)
    kx = torch.randperm(n0, generator=cpu)[:KB].sort().values
    kz = torch.randperm(n2, generator=cpu)[:KB].sort().values
    yl = n1 // P
    ky = torch.cat([d * yl + torch.randperm(yl, generator=cpu)[:KB // P] for d in range(P)])  # two rows of every device's Y-slab

    def phases(n, k, lo, cnt):  # [cnt, KB] fp64 table of exp(-2 pi i k x / n), the exponent reduced in integers first
        x = torch.arange(lo, lo + cnt, dtype=torch.int64)[:, None]
        ph = (-2.0 * math.pi / n) * ((x * k[None, :].to(torch.int64)) % n).to(torch.float64)
        return torch.complex(torch.cos(ph), torch.sin(ph)).to(gpu)

    comm = api.Comm.local(P)
    ins, outs, plans = [], [], []
    want = torch.zeros(KB, KB, KB, dtype=torch.complex128, device=gpu)  # [kx][ky][kz]
    wy, wz = phases(n1, ky, 0, n1), phases(n2, kz, 0, n2)
    for g in range(P):
        x0, xs = _slab(n0, P, g)
        cnt = xs * n1 * n2
        a = torch.empty(cnt, dtype=torch.complex64, device=gpu)
        av = torch.view_as_real(a)
        av.copy_(torch.rand(cnt, 2, generator=gen, device=gpu, dtype=torch.float32))  # uniform [0, 1) like heFFTe's worlds
        v = a.view(xs, n1, n2)
        wx = phases(n0, kx, x0, xs)                                       # [xs, KB]
        step = 16
        for c0 in range(0, xs, step):
            t1 = v[c0:c0 + step].to(torch.complex128) @ wz                # [step, n1, KB(kz)]
            t2 = torch.einsum("xyk,yj->xjk", t1, wy)                      # [step, KB(ky), KB(kz)]
            want += torch.einsum("xi,xjk->ijk", wx[c0:c0 + step], t2)
            del t1, t2
        b = torch.zeros(cnt, dtype=torch.complex64, device=gpu)
        ins.append(a)
        outs.append(b)
        plans.append(api.Plan(n0, n1, n2, a, b, comm, g, P, api.FORWARD,
                              api.PLAN_INPUT_FROM_IN | (api.PLAN_OVERLAP if overlap else 0)))
    _run(plans)
    got = torch.zeros_like(want)
    for j, kyj in enumerate(ky.tolist()):
        d, yy = kyj // yl, kyj % yl
        o = outs[d].view(yl, n2, n0)                                      # [yy][z][kx]
        got[:, j, :] = o[yy][kz.to(gpu)][:, kx.to(gpu)].t().to(torch.complex128)
    for p in plans:
        p.destroy()
    comm.destroy()
    scale = float(want.abs().max().item())
    worst = float((got - want).abs().max().item())
    assert worst / scale <= 5e-4, f"{KB ** 3} bins: max |diff| / max |ref| = {worst / scale:.3e}"
    # the bins are not trivially small: a uniform [0, 1) world has a large mean (bin 0) and ~sqrt(N) elsewhere
    assert scale > 1e4
