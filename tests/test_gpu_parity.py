"""-m gpu parity tests: the HIP path (through the C-ABI) against the oracle on the same seeded inputs.

Tolerance: the north star asks for <= 1e-11 max error in fp64 (heFFTe's bar, test_common.h:136-140); it is applied to
the forward result relative to max|reference| ("parity metric").  fp32: 5e-4 (same heFFTe table).
"""
import os
import threading

import numpy as np
import pytest

from oracle import slab_oracle as so

pytestmark = pytest.mark.gpu

TOL = {"f64": 1e-11, "f32": 5e-4}
# tuned plans (dfft_plans.h) ...
TUNED = [2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 16, 24, 25, 32, 40, 48, 49, 64, 80, 96, 100, 125, 128, 160, 192, 200, 256, 320, 343,
         384, 400, 512, 640, 768, 1000, 1024, 1280, 1536, 2048, 27, 81, 243, 625, 729, 2187, 3125, 2401, 4096]
# ... and 7-smooth lengths served by the run-time-scheduled kernel (dfft_generic.hip): every radix mix, up to 4096
GENERIC = [15, 18, 20, 21, 35, 36, 45, 50, 60, 63, 72, 90, 105, 120, 144, 210, 240, 250, 360, 500, 720,
           800, 2000, 3072, 4000, 3600]
LENGTHS = TUNED + GENERIC


def _torch_dtype(name):
    import torch
    return torch.complex128 if name == "f64" else torch.complex64


def _rel_err(got, ref):
    return float(np.max(np.abs(got - ref)) / max(np.max(np.abs(ref)), 1e-300))


@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize("n", LENGTHS)
def test_fft1d_rows_vs_oracle(gpu, n, prec):
    import torch
    from distributedfft_amd import api
    rng = np.random.default_rng(n)
    batch = 37 if n <= 256 else 9
    x = (rng.uniform(-1, 1, (batch, n)) + 1j * rng.uniform(-1, 1, (batch, n)))
    xt = torch.from_numpy(x).to(gpu).to(_torch_dtype(prec))
    ref_f = so.c_fft1d(x, +1)           # C restatement
    ref_np = np.fft.fft(x)              # independent
    assert _rel_err(ref_f, ref_np) < 1e-13
    got = api.fft1d_rows(xt, api.FORWARD).cpu().numpy()
    assert _rel_err(got, ref_f) < TOL[prec], f"rows n={n}"
    back = api.fft1d_rows(xt, api.BACKWARD).cpu().numpy()
    assert _rel_err(back, np.fft.ifft(x) * n) < TOL[prec]
    # in place
    y = xt.clone()
    api.fft1d_rows(y, api.FORWARD, out=y)
    assert _rel_err(y.cpu().numpy(), ref_f) < TOL[prec]


# plane shapes of the reference's published 2D table (templateFFT/csv/batch_result2D.csv: powers of two 128 ... 2048, 120 x 120, 240 x 100,
# 245 x 245, 360 x 360, 243 x 243, 729 x 243, 625 x 125, 343 x 343), the shapes of the one-launch stage (256 / 512 / 768-point axes) with plane
# counts that make one, two and several Infinity-Cache phases, odd / ragged widths, a four-step axis
@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize("n1,n2,batch", [(512, 512, 3), (256, 256, 70), (512, 256, 5), (256, 512, 131), (768, 512, 40), (512, 512, 70),
                                         (2048, 2048, 2), (1024, 512, 3), (128, 2048, 4), (2048, 128, 4), (120, 120, 9), (100, 240, 5),
                                         (245, 245, 3), (360, 360, 2), (243, 243, 4), (243, 729, 2), (125, 625, 3), (343, 343, 2),
                                         (64, 21, 7), (7, 8, 5), (2, 16, 3), (16, 2, 3), (8192, 8, 2)])
def test_fft2d_batch_vs_numpy(gpu, n1, n2, batch, prec):
    """dfft_fft2d_batch (the plan's t0 as an entry point: Infinity-Cache chunks, one-launch stage where built) against np.fft.fft2,
    forward and backward, out of place (input untouched) and in place."""
    import torch
    from distributedfft_amd import api
    rng = np.random.default_rng(n1 * 4099 + n2)
    x = (rng.uniform(-1, 1, (batch, n1, n2)) + 1j * rng.uniform(-1, 1, (batch, n1, n2)))
    xt = torch.from_numpy(x).to(gpu).to(_torch_dtype(prec))
    keep = xt.clone()
    ref = np.fft.fft2(x, axes=(1, 2))
    got = api.fft2d_batch(xt, api.FORWARD).cpu().numpy()
    assert _rel_err(got, ref) < TOL[prec], f"2D {n1}x{n2}x{batch}"
    assert torch.equal(xt, keep)  # out of place: the input is left alone
    back = api.fft2d_batch(xt, api.BACKWARD).cpu().numpy()
    assert _rel_err(back, np.fft.ifft2(x, axes=(1, 2)) * (n1 * n2)) < TOL[prec]
    assert torch.equal(xt, keep)
    y = xt.clone()
    api.fft2d_batch(y, api.FORWARD, out=y)
    assert _rel_err(y.cpu().numpy(), ref) < TOL[prec]
    api.fft2d_batch(y, api.BACKWARD, out=y)  # round trip in place
    assert _rel_err(y.cpu().numpy() / (n1 * n2), x) < TOL[prec] * 10


def test_fft2d_batch_is_the_plans_t0(gpu):
    """Same planes through dfft_fft2d_batch and through two whole-buffer 1-D passes: the one-launch form within 1e-14 of the 1-D
    entry points (lazy publish: last bit or two), calls repeat bit for bit, dfft_trim() drops the cached control blocks."""
    import torch
    from distributedfft_amd import _lib, api
    g = torch.Generator(device=gpu)
    g.manual_seed(7)
    x = (torch.rand(96, 512, 512, generator=g, device=gpu, dtype=torch.float64) - 0.5).to(torch.complex128)
    a = api.fft2d_batch(x)
    b = api.fft2d_batch(x)
    assert torch.equal(a, b)
    c = api.fft1d_cols(api.fft1d_rows(x.reshape(-1, 512)).reshape(96, 512, 512))
    assert float((a - c).abs().max() / c.abs().max()) < 1e-14
    assert _lib.load().dfft_trim() == 0
    assert torch.equal(api.fft2d_batch(x), a)


@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize("n", LENGTHS)
# full tiles / ragged last tile with an odd column count (GENERAL variant; fp32 cannot pair columns) / ragged, even
# (fp32: GENERAL variant of the column-pair kernel)
@pytest.mark.parametrize("width", [32, 21, 20])
def test_fft1d_cols_vs_oracle(gpu, n, width, prec):
    import torch
    from distributedfft_amd import api
    rng = np.random.default_rng(1000 + n)
    batch = 3
    x = (rng.uniform(-1, 1, (batch, n, width)) + 1j * rng.uniform(-1, 1, (batch, n, width)))
    xt = torch.from_numpy(x).to(gpu).to(_torch_dtype(prec))
    ref = np.fft.fft(x, axis=1)
    got = api.fft1d_cols(xt, api.FORWARD).cpu().numpy()
    assert _rel_err(got, ref) < TOL[prec], f"cols n={n} width={width}"
    back = api.fft1d_cols(xt, api.BACKWARD).cpu().numpy()
    assert _rel_err(back, np.fft.ifft(x, axis=1) * n) < TOL[prec]


def _run_plans(gpu, N, P, prec, x, direction, flags=0, inputs=None, exec_flags=0, describes=None):
    """Create P plans (virtual devices on one GPU, LOCAL communicator), execute them from P threads, return outputs
    (describes: a list that receives dfft_plan_describe of every plan)."""
    import torch
    from distributedfft_amd import api
    n0, n1, n2 = N
    comm = api.Comm.local(P) if P > 1 else None
    tdt = _torch_dtype(prec)
    plans, ins, outs = [], [], []
    for g in range(P):
        mc = api.get_max_data_count(n0, n1, n2, P, g == P - 1)
        a = torch.zeros(mc, dtype=tdt, device=gpu)
        b = torch.zeros(mc, dtype=tdt, device=gpu)
        src = torch.from_numpy(np.ascontiguousarray(inputs[g]).reshape(-1)).to(gpu).to(tdt)
        a[:src.numel()] = src
        torch.cuda.synchronize()
        plans.append(api.Plan(n0, n1, n2, a, b, comm, g, P, direction, flags))
        if describes is not None:
            describes.append(plans[-1].describe())
        ins.append(a)
        outs.append(b)
    errs = []

    def work(g):
        try:
            plans[g].execute(exec_flags)
            plans[g].sync()
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=work, args=(g,)) for g in range(P)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    res = [o.cpu().numpy() for o in outs]
    times = [p.stage_times() for p in plans]
    for p in plans:
        p.destroy()
    if comm:
        comm.destroy()
    return res, times


SHAPES = [
    ((8, 8, 8), 1), ((16, 12, 10), 1), ((2, 3, 4), 1), ((64, 64, 64), 1), ((32, 48, 24), 1), ((128, 96, 64), 1),
    ((64, 64, 64), 2), ((64, 64, 64), 4), ((32, 48, 24), 2), ((128, 128, 32), 8),
    ((10, 10, 8), 4),     # uneven in X and Y: xl=3 (last 1), yl=3 (last 1)
    ((25, 10, 16), 4),    # uneven X (7,7,7,4) and Y (3,3,3,1)
    ((24, 10, 12), 4),    # ragged N2 (12 % 8 != 0, 12 % 16 != 0) with uneven Y (3,3,3,1)
    ((48, 100, 12), 2),   # radix-5 Y axis, ragged Z
    ((14, 49, 16), 2),    # radix-7 X and Y axes
    ((343, 8, 8), 1),     # 7*7*7 X axis
    # long X / Y axes: staged transposed store with 16 points per thread (1024), half-line tiles (2048; fp32: staged
    # column pairs), 24 points per thread (768), and the 512-point headline kernels on a small slab
    ((2048, 4, 16), 1), ((1024, 6, 32), 2), ((768, 4, 16), 1), ((8, 2048, 16), 1), ((512, 8, 32), 1),
    # 2048-point X pass: paired half-line tiles (fft_dual_tiles_kernel) over per-peer blocks of the exchange buffer (P = 2, 4),
    # and a column count that is not a multiple of a tile pair (single 4-column tiles)
    ((2048, 4, 16), 2), ((2048, 8, 8), 4), ((2048, 3, 4), 1),
    # 2048-point Y pass: DIF-split full-line tiles (fft_dif2_tiles_kernel) on natural maps (P = 1) and on the packed exchange
    # layout (P = 2, 4: blocks of 1024 / 512 rows, both directions); 24 columns = 12 fp32 pairs fall back to half-line tiles
    ((4, 2048, 24), 1), ((8, 2048, 16), 2), ((16, 2048, 8), 4),
    # lengths without a tuned plan (run-time-scheduled kernel) on every axis, with pack / transposed store / uneven slabs
    ((20, 36, 40), 1), ((20, 36, 40), 4), ((45, 50, 18), 4), ((1000, 6, 8), 2), ((8, 640, 12), 2), ((4096, 2, 8), 1),
    ((16, 24, 1536), 2), ((60, 64, 20), 8),
    # the top of the single-pass range on every axis: 4096 (16 points per thread, 2-column tiles) and 2401 = 7^4
    ((4, 4096, 8), 2), ((2, 8, 4096), 1), ((2401, 3, 4), 1), ((4, 6, 2401), 2),
    # round 6, the "lean" lengths (1000 / 1280 / 1536 points: half-line tiles wherever a variant needs per-point offsets): X pass staged
    # on half-line tiles (fp64 and pairs), Y pass into the packed layout, uneven Y splits (ragged address terms), P = 1 plain twins
    ((1536, 8, 16), 1), ((1536, 8, 16), 2), ((1280, 8, 16), 4), ((1280, 4, 16), 1), ((1000, 8, 16), 1), ((1000, 8, 16), 2),
    ((8, 1536, 16), 2), ((4, 1536, 16), 1), ((8, 1280, 16), 4), ((8, 1000, 16), 2), ((12, 1000, 8), 3), ((12, 1280, 8), 3), ((12, 1536, 12), 4),
    ((729, 9, 8), 3), ((8, 729, 8), 2),
    # round 6, an ODD Z length: fp32 column launches cannot run on column pairs and take the scalar float2 fall-back (launch_scalar32,
    # eight kernels per length on a geometry of its own) -- short and long X / Y axes, packed and ragged
    ((512, 8, 9), 1), ((8, 1024, 7), 2), ((2048, 2, 5), 1), ((4, 768, 5), 2), ((640, 4, 3), 2), ((16, 2048, 3), 2), ((1536, 4, 3), 1),
    ((12, 400, 5), 3), ((343, 4, 7), 1),
]


@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize("flags", [0, 1], ids=["fused", "unfused"])
@pytest.mark.parametrize("N,P", SHAPES)
def test_slab_forward_backward_vs_oracle(gpu, N, P, prec, flags):
    n0, n1, n2 = N
    if so.slab_size(n1, P, P - 1) < 1 or so.slab_size(n0, P, P - 1) < 1:
        pytest.skip("decomposition leaves the last device empty (the reference cannot run it either)")
    x = so.random_input(N, seed=n0 * 10007 + n1 * 101 + n2)
    ref = so.fftn_reference(x, P)
    flat, _ = so.c_slab_fft3d(x, N, P, +1)
    cref = so.split_forward_output(flat, N, P)
    inputs = [x[so.slab_start(n0, P, g):so.slab_start(n0, P, g) + so.slab_size(n0, P, g)] for g in range(P)]
    outs, _ = _run_plans(gpu, N, P, prec, x, +1, flags, inputs)
    scale = max(np.abs(r).max() for r in ref)
    for d in range(P):
        cnt = ref[d].size
        got = outs[d][:cnt].reshape(ref[d].shape)
        assert np.abs(cref[d] - ref[d]).max() / scale < 1e-12
        assert np.abs(got - cref[d]).max() / scale < TOL[prec], f"forward N={N} P={P} dev={d}"
    # backward consumes the forward layout and returns the input slabs times N0*N1*N2
    bouts, _ = _run_plans(gpu, N, P, prec, None, -1, flags, [r for r in ref])
    for g in range(P):
        cnt = inputs[g].size
        got = bouts[g][:cnt].reshape(inputs[g].shape) / float(n0 * n1 * n2)
        assert np.abs(got - inputs[g]).max() < TOL[prec] * 10, f"backward N={N} P={P} dev={g}"


def test_driver_input_roundtrip_error_metric(gpu):
    """The reference driver's own self-check (fftSpeed3d_c2c.cpp:56-91) on its own input at 64^3, P = 1 and 4."""
    N = (64, 64, 64)
    for P in (1, 4):
        inputs = [so.driver_input(N, P, g) for g in range(P)]
        full = np.concatenate(inputs, axis=0)
        ref = so.fftn_reference(full, P)
        outs, _ = _run_plans(gpu, N, P, "f64", None, +1, 0, inputs)
        scale = max(np.abs(r).max() for r in ref)
        for d in range(P):
            got = outs[d][:ref[d].size].reshape(ref[d].shape)
            assert np.abs(got - ref[d]).max() / scale < 1e-11
        fwd = [outs[d][:ref[d].size].reshape(ref[d].shape) for d in range(P)]
        back, _ = _run_plans(gpu, N, P, "f64", None, -1, 0, fwd)
        err = max(so.driver_error(inputs[g].reshape(-1), back[g][:inputs[g].size], N) for g in range(P))
        assert err < 1e-11  # the reference's README prints 4.2e-15 for 512^3 (README.md:55)


def test_input_from_in_flag_repeats_identically(gpu):
    import torch
    from distributedfft_amd import api
    N = (64, 32, 32)
    x = so.random_input(N, seed=5)
    ref = so.fftn_reference(x, 1)[0]
    a = torch.from_numpy(x.reshape(-1)).to(gpu)
    b = torch.zeros_like(a)
    torch.cuda.synchronize()
    plan = api.Plan(*N, a, b, None, 0, 1, api.FORWARD, api.PLAN_INPUT_FROM_IN)
    for _ in range(3):
        plan.execute()
    plan.sync()
    got = b.cpu().numpy().reshape(ref.shape)
    assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-11
    assert np.array_equal(a.cpu().numpy().reshape(N), x)  # input untouched
    t = plan.stage_times()
    assert all(v >= 0 for v in t) and t[0] > 0 and t[3] > 0
    plan.destroy()


def test_rccl_single_rank_communicator(gpu):
    """RCCL path at world size 1 (the only size one GPU allows): communicator creation + self chunk."""
    import torch
    from distributedfft_amd import api
    uid = api.Comm.rccl_unique_id()
    comm = api.Comm.rccl(uid, 1, 0)
    N = (32, 32, 32)
    x = so.random_input(N, seed=9)
    a = torch.from_numpy(x.reshape(-1)).to(gpu)
    b = torch.zeros_like(a)
    torch.cuda.synchronize()
    plan = api.Plan(*N, a, b, comm, 0, 1, api.FORWARD)
    plan.execute()
    plan.sync()
    ref = so.fftn_reference(x, 1)[0]
    assert np.abs(b.cpu().numpy().reshape(ref.shape) - ref).max() / np.abs(ref).max() < 1e-11
    plan.destroy()
    comm.destroy()


def test_execute_without_stage_events(gpu):
    """DFFT_EXEC_NO_TIMING: same result, no stage-boundary events; asking for stage times afterwards is an error, and
    the next timed execute brings them back."""
    import torch
    from distributedfft_amd import api
    N = (32, 16, 24)
    x = so.random_input(N, seed=31)
    ref = so.fftn_reference(x, 1)[0]
    a = torch.from_numpy(x.reshape(-1)).to(gpu)
    b = torch.zeros_like(a)
    torch.cuda.synchronize()
    plan = api.Plan(*N, a, b, None, 0, 1, api.FORWARD, api.PLAN_INPUT_FROM_IN)
    plan.execute(api.EXEC_NO_TIMING)
    plan.sync()
    assert np.abs(b.cpu().numpy().reshape(ref.shape) - ref).max() / np.abs(ref).max() < 1e-11
    with pytest.raises(api.DfftError):
        plan.stage_times()
    plan.execute()
    assert len(plan.stage_times()) == 4
    plan.destroy()


def test_untimed_executes_back_to_back_are_bit_identical(gpu):
    """Un-timed executes (DFFT_EXEC_NO_TIMING, what a production loop and bench.py's timed steps use) queued back to back on
    the plan's stream: same results as the event-timed executes, input changes are seen, a changed scale factor takes effect,
    both directions, a chunked and a cache-resident size."""
    import torch
    from distributedfft_amd import api
    for N in ((64, 32, 48), (512, 256, 256)):
        n = N[0] * N[1] * N[2]
        g = torch.Generator(device=gpu)
        g.manual_seed(11)
        a = torch.complex(torch.rand(n, generator=g, device=gpu, dtype=torch.float64) - 0.5,
                          torch.rand(n, generator=g, device=gpu, dtype=torch.float64) - 0.5)
        b, b_ref, c = torch.zeros_like(a), torch.zeros_like(a), torch.zeros_like(a)
        ref = api.Plan(*N, a, b_ref, None, 0, 1, api.FORWARD, api.PLAN_INPUT_FROM_IN)
        p = api.Plan(*N, a, b, None, 0, 1, api.FORWARD, api.PLAN_INPUT_FROM_IN)
        q = api.Plan(*N, b, c, None, 0, 1, api.BACKWARD, api.PLAN_INPUT_FROM_IN)
        for it in range(4):
            a.mul_(1.0 + 0.125 * it)       # new input every round
            torch.cuda.synchronize()
            ref.execute()
            ref.sync()
            for _ in range(3):
                p.execute(api.EXEC_NO_TIMING)
            p.sync()
            assert torch.equal(b, b_ref), (N, it)
            q.execute(api.EXEC_NO_TIMING)
            q.sync()
            assert (c / n - a).abs().max().item() < 1e-12
        p.set_scale(0.5)
        ref.set_scale(0.5)
        ref.execute()
        ref.sync()
        p.execute(api.EXEC_NO_TIMING)
        p.sync()
        assert torch.equal(b, b_ref)
        for pl in (ref, p, q):
            pl.destroy()


@pytest.mark.parametrize("N,chunk", [((16, 256, 256), 5), ((12, 512, 256), 4), ((9, 256, 512), 2), ((70, 512, 512), 64), ((512, 256, 256), 0),
                                     ((10, 768, 512), 3), ((64, 768, 512), 0)])   # 768-point Y axis: config 4's planes (round 5)
def test_one_launch_t0_is_bit_identical(gpu, N, chunk, monkeypatch):
    """t0 as ONE persistent launch (dfft_zy.hip: ticket-ordered row and column units, sc1 hand-off through the hand-over buffer)
    against the two launches per cache chunk it replaces: with several chunks and a ragged last chunk, in both directions, also
    when executes are queued back to back -- the eager-publish kernel bit-identical, the lazy-publish one (the default) equal
    to the last bit or two and deterministic; and both against the oracle."""
    import torch
    from distributedfft_amd import api
    monkeypatch.setenv("DFFT_PAD", "1")            # small slabs get the hand-over buffer too
    if chunk:
        monkeypatch.setenv("DFFT_CHUNK_PLANES", str(chunk))
    n = N[0] * N[1] * N[2]
    x = so.random_input(N, seed=N[0] + 3)
    a = torch.from_numpy(x.reshape(-1)).to(gpu)
    outs = {}
    for mode in ("0", "1", "lazy", "lazy2"):  # "1": the eager-publish kernel (DFFT_ZY_LAZY=0); "lazy": the default one, twice
        monkeypatch.setenv("DFFT_T0_ONE_LAUNCH", "0" if mode == "0" else "1")
        monkeypatch.setenv("DFFT_ZY_LAZY", "1" if mode.startswith("lazy") else "0")
        b, c = torch.zeros_like(a), torch.zeros_like(a)
        p = api.Plan(*N, a, b, None, 0, 1, api.FORWARD, api.PLAN_INPUT_FROM_IN)
        q = api.Plan(*N, b, c, None, 0, 1, api.BACKWARD, api.PLAN_INPUT_FROM_IN)
        assert ("yz_stage=one-launch-lazy" in p.describe()) == mode.startswith("lazy"), p.describe()
        for _ in range(3):
            p.execute(api.EXEC_NO_TIMING)
        p.execute()
        p.sync()
        assert len(p.stage_times()) == 4
        for _ in range(2):
            q.execute(api.EXEC_NO_TIMING)
        q.sync()
        outs[mode] = (b.clone(), c.clone())
        p.destroy()
        q.destroy()
    assert torch.equal(outs["0"][0], outs["1"][0]) and torch.equal(outs["0"][1], outs["1"][1])
    # The lazy-publish kernel keeps the products of its twiddle powers out of the unit loop's invariants, and the compiler then
    # contracts a few multiply-adds the other way round: its results differ from the other two forms in the last bit (3e-16 of
    # max|X| forward, 1.3e-15 after the inverse; profiles/r03/experiments/lazy_vs_eager_bits.log), are just as close to the
    # reference, and are the same from run to run and from plan to plan.
    assert torch.equal(outs["lazy"][0], outs["lazy2"][0]) and torch.equal(outs["lazy"][1], outs["lazy2"][1])
    for k in (0, 1):
        assert ((outs["lazy"][k] - outs["0"][k]).abs().max() / outs["0"][k].abs().max()).item() < 1e-14
    ref = so.fftn_reference(x, 1)[0]
    for mode in ("1", "lazy"):
        got = outs[mode][0].cpu().numpy().reshape(ref.shape)
        assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-11
        assert (outs[mode][1] / n - a).abs().max().item() < 1e-11


@pytest.mark.parametrize("N,prec,chunk", [((16, 256, 256), "f32", 5), ((12, 512, 128), "f32", 4), ((9, 1024, 64), "f64", 2), ((10, 96, 2048), "f64", 3),
                                          ((7, 768, 256), "f32", 3)])
def test_inverse_chunk_loop_rows_first_vs_oracle(gpu, N, prec, chunk, monkeypatch):
    """Single-GPU backward plans that cut their slab into cache chunks on a hand-over buffer run the inverse YZ stage Z rows first
    (hand-over buffer -> result buffer, then Y columns in place on the cache-resident chunk; round 6, dfft_plan.cpp execute_backward).
    Shapes the one-launch stage does not take (fp32, or a 1024- / 2048-point fp64 axis), several chunks with a ragged last one: the
    backward transform of the forward result against numpy's inverse and against the input (round trip), and against the
    columns-first order (DFFT_ZY_INV_ROWS_FIRST=0) to the last bits."""
    import torch
    from distributedfft_amd import api
    monkeypatch.setenv("DFFT_PAD", "1")            # small slabs get the hand-over buffer too
    monkeypatch.setenv("DFFT_CHUNK_PLANES", str(chunk))
    n = N[0] * N[1] * N[2]
    cdt, tol = (np.complex128, 1e-11) if prec == "f64" else (np.complex64, 5e-4)
    x = so.random_input(N, seed=N[1] + 7).astype(cdt)
    a = torch.from_numpy(x.reshape(-1)).to(gpu)
    outs = {}
    for order in ("rows", "cols"):
        if order == "cols":
            monkeypatch.setenv("DFFT_ZY_INV_ROWS_FIRST", "0")
        b, c = torch.zeros_like(a), torch.zeros_like(a)
        p = api.Plan(*N, a, b, None, 0, 1, api.FORWARD, api.PLAN_INPUT_FROM_IN)
        q = api.Plan(*N, b, c, None, 0, 1, api.BACKWARD, api.PLAN_INPUT_FROM_IN)
        assert "yz_stage=two-launches-per-chunk" in q.describe() and "handover=padded-buffer" in q.describe(), q.describe()
        p.execute()
        p.sync()
        for _ in range(2):
            q.execute(api.EXEC_NO_TIMING)
        q.sync()
        outs[order] = c.clone()
        fw = b.cpu().numpy().reshape(N[1], N[2], N[0])
        p.destroy()
        q.destroy()
    exp = np.fft.ifftn(np.transpose(fw, (2, 0, 1)).astype(np.complex128)) * n      # the unnormalised inverse, [x][y][z]
    for order in ("rows", "cols"):
        got = outs[order].cpu().numpy().reshape(N)
        assert np.abs(got - exp).max() / np.abs(exp).max() < tol, order
        assert np.abs(got / n - x).max() / np.abs(x).max() < tol, order
    assert ((outs["rows"] - outs["cols"]).abs().max() / outs["cols"].abs().max()).item() < (1e-14 if prec == "f64" else 1e-6)


@pytest.mark.parametrize("direction", [+1, -1])
def test_one_launch_t0_failure_is_loud_and_recovered(gpu, direction, monkeypatch):
    """A one-launch YZ stage that gives up must never hand back garbage with DFFT_OK.  DFFT_ZY_FAULT=n makes launch n
    of a plan's stage wait for producers that never come; its consumers exhaust their (here: few) polls, the kernel raises the
    sticky error word and the pinned host word.  (a) a host-synchronised execute -- what the reference-named wrapper does -- sees
    it, runs the transform again on two launches per chunk and returns the right answer; (b) an asynchronous execute reports it
    at dfft_plan_sync, the executes queued behind the failed launch refuse to run instead of decoding a wrapped ticket range, the
    next execute call reports nothing new and computes correctly on the two-launch stage."""
    import torch
    from distributedfft_amd import api
    from distributedfft_amd._lib import DfftError
    monkeypatch.setenv("DFFT_PAD", "1")
    monkeypatch.setenv("DFFT_CHUNK_PLANES", "4")
    monkeypatch.setenv("DFFT_T0_ONE_LAUNCH", "1")
    monkeypatch.setenv("DFFT_ZY_SPIN_POLLS", "2000")
    N = (12, 512, 512)
    n = N[0] * N[1] * N[2]
    x = so.random_input(N, seed=5)
    a = torch.from_numpy(x.reshape(-1)).to(gpu)
    # reference result from a healthy two-launch plan
    monkeypatch.setenv("DFFT_T0_ONE_LAUNCH", "0")
    good = torch.zeros_like(a)
    r = api.Plan(*N, a, good, None, 0, 1, direction, api.PLAN_INPUT_FROM_IN)
    r.execute(api.EXEC_NO_TIMING)
    r.sync()
    r.destroy()
    monkeypatch.setenv("DFFT_T0_ONE_LAUNCH", "1")
    # (a) host-synchronised execute: second launch of the stage fails, the execute recovers by itself
    monkeypatch.setenv("DFFT_ZY_FAULT", "2")
    b = torch.zeros_like(a)
    p = api.Plan(*N, a, b, None, 0, 1, direction, api.PLAN_INPUT_FROM_IN)
    assert "yz_stage=one-launch" in p.describe()
    p.execute(api.EXEC_SYNC_STAGES)
    assert "yz_stage=one-launch" in p.describe() and (b - good).abs().max().item() <= 1e-14 * good.abs().max().item()
    b.zero_()
    p.execute(api.EXEC_SYNC_STAGES)          # the faulty launch
    assert "yz_stage=two-launches-per-chunk" in p.describe(), p.describe()
    assert torch.equal(b, good)
    b.zero_()
    p.execute(api.EXEC_NO_TIMING)
    p.sync()
    assert torch.equal(b, good)
    p.destroy()
    # (b) asynchronous executes: the failure surfaces at the next synchronisation through the library
    monkeypatch.setenv("DFFT_ZY_FAULT", "1")
    p = api.Plan(*N, a, b, None, 0, 1, direction, api.PLAN_INPUT_FROM_IN)
    # the first launch fails, two more are queued behind it.  dfft_execute looks at the pinned error word on entry, and with so
    # few polls the failing kernel gives up within milliseconds: on a slow host the second or third execute call may already
    # report the failure -- wherever it surfaces, it must be the same loud error, once
    with pytest.raises(DfftError, match="one-launch YZ stage gave up"):
        for _ in range(3):
            p.execute(api.EXEC_NO_TIMING)
        p.sync()
    p.sync()                                  # (drains whatever was queued before the report; nothing new to report)
    assert "yz_stage=two-launches-per-chunk" in p.describe()
    b.zero_()
    p.execute(api.EXEC_NO_TIMING)
    p.sync()
    assert torch.equal(b, good)
    p.destroy()
    # (c) nobody synchronises through the library: the next execute call reports it before queueing anything
    p = api.Plan(*N, a, b, None, 0, 1, direction, api.PLAN_INPUT_FROM_IN)
    p.execute(api.EXEC_NO_TIMING)
    torch.cuda.synchronize()
    with pytest.raises(DfftError, match="one-launch YZ stage gave up"):
        p.execute(api.EXEC_NO_TIMING)
    b.zero_()
    p.execute(api.EXEC_NO_TIMING)
    p.sync()
    assert torch.equal(b, good)
    p.destroy()


def test_one_launch_t0_failure_is_collective(gpu, monkeypatch):
    """P > 1: the one-launch YZ stage of ONE device gives up.  Its garbage has gone through the exchange by the time the
    host sees the error word, so every device must return the error from that host-synchronised execute -- not only the one whose
    stage failed -- and all of them continue on two launches per chunk with correct results.  An asynchronous execute on the
    failing device still queues its exchange (its peers must not be left waiting) and reports afterwards."""
    import torch
    from distributedfft_amd import api
    from distributedfft_amd._lib import DfftError
    N, P = (8, 512, 512), 2
    n0, n1, n2 = N
    x = so.random_input(N, seed=77)
    ref = so.fftn_reference(x, P)
    scale = max(np.abs(r).max() for r in ref)
    monkeypatch.setenv("DFFT_ZY_SPIN_POLLS", "2000")
    for first_exec in (api.EXEC_SYNC_STAGES, api.EXEC_NO_TIMING):
        comm = api.Comm.local(P)
        plans, outs = [], []
        for g in range(P):
            mc = api.get_max_data_count(n0, n1, n2, P, g == P - 1)
            a = torch.zeros(mc, dtype=torch.complex128, device=gpu)
            b = torch.zeros(mc, dtype=torch.complex128, device=gpu)
            src = torch.from_numpy(np.ascontiguousarray(x[g * (n0 // P):(g + 1) * (n0 // P)]).reshape(-1)).to(gpu)
            a[:src.numel()] = src
            torch.cuda.synchronize()
            if g == 1:
                monkeypatch.setenv("DFFT_ZY_FAULT", "1")   # only this device's first launch of the stage fails
            plans.append(api.Plan(n0, n1, n2, a, b, comm, g, P, api.FORWARD, api.PLAN_INPUT_FROM_IN))
            monkeypatch.delenv("DFFT_ZY_FAULT", raising=False)
            assert "yz_stage=one-launch" in plans[-1].describe()
            outs.append((a, b))
        seen = [None] * P

        def work(g):
            try:
                plans[g].execute(first_exec)
                if first_exec == api.EXEC_NO_TIMING:      # asynchronous: the failing device reports at its next sync / execute
                    plans[g].sync()
            except DfftError as e:
                seen[g] = str(e)

        th = [threading.Thread(target=work, args=(g,)) for g in range(P)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert seen[1] and "one-launch YZ stage gave up" in seen[1], seen
        if first_exec == api.EXEC_SYNC_STAGES:             # host-synchronised: every device knows
            assert seen[0] and "one-launch YZ stage gave up" in seen[0], seen
            assert all("yz_stage=two-launches-per-chunk" in p.describe() for p in plans)
        seen = [None] * P
        first_exec_again = api.EXEC_SYNC_STAGES

        def again(g):
            try:
                plans[g].execute(first_exec_again)
            except DfftError as e:
                seen[g] = str(e)

        th = [threading.Thread(target=again, args=(g,)) for g in range(P)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert seen == [None] * P, seen
        for g in range(P):
            got = outs[g][1].cpu().numpy()[:ref[g].size].reshape(ref[g].shape)
            assert np.abs(got - ref[g]).max() / scale < 1e-11, g
        for p in plans:
            p.destroy()
        comm.destroy()


@pytest.mark.parametrize("rot", ["0", "1"])
@pytest.mark.parametrize("N,P", [((8, 256, 256), 2), ((16, 256, 512), 4), ((16, 512, 256), 2), ((32, 256, 256), 8), ((24, 512, 512), 4),
                                 ((16, 768, 512), 8), ((8, 768, 512), 2),    # config 4's planes: blocks of 96 / 384 rows per destination
                                 ((16, 768, 512), 4)])   # ... and 192 rows: two Y sub-blocks of 96 in the overlapped plan (one launch for all parts)
def test_one_launch_t0_with_exchange_is_bit_identical(gpu, N, P, rot, monkeypatch):
    """P > 1: the one-launch YZ stage stores its column results straight into the packed (and, with DFFT_ROT=1, row-rotated) send
    layout, the inverse reads the packed receive layout -- whole slabs in the serial pipeline, X-plane parts in the overlapped
    one.  The eager-publish kernel is bit-identical to two launches per chunk in both directions; the lazy-publish one (the
    default since round 4) agrees to the last bit or two, is the same in the serial and the overlapped pipeline, and both are
    checked against the oracle."""
    from distributedfft_amd import api
    monkeypatch.setenv("DFFT_ROT", rot)
    monkeypatch.setenv("DFFT_CHUNK_PLANES", "3")   # several phases per slab / part
    n0, n1, n2 = N
    x = so.random_input(N, seed=77 + P)
    ref = so.fftn_reference(x, P)
    inputs = [x[so.slab_start(n0, P, g):so.slab_start(n0, P, g) + so.slab_size(n0, P, g)] for g in range(P)]
    res, one = {}, {}
    all_flags = (api.PLAN_INPUT_FROM_IN, api.PLAN_INPUT_FROM_IN | api.PLAN_OVERLAP)
    for mode in ("0", "eager", "lazy"):
        monkeypatch.setenv("DFFT_T0_ONE_LAUNCH", "0" if mode == "0" else "all")   # planes with a 256-point axis use the stage only on request
        monkeypatch.setenv("DFFT_ZY_LAZY", "0" if mode == "eager" else "1")
        for flags in all_flags:
            desc = []
            fwd, _ = _run_plans(gpu, N, P, "f64", x, +1, flags, inputs, describes=desc)
            bwd, _ = _run_plans(gpu, N, P, "f64", None, -1, flags, [r for r in ref], describes=desc)
            res[(mode, flags)] = (fwd, bwd)
            # (an overlapped plan whose Y sub-blocks are narrower than the column unit's stride keeps two launches per chunk)
            one[(mode, flags)] = all("yz_stage=one-launch" in t for t in desc)
            if mode == "0":
                assert not any("yz_stage=one-launch" in t for t in desc)
            elif flags == all_flags[0]:   # the serial pipeline of every shape in the list runs the stage: the test is not vacuous
                assert one[(mode, flags)] and all(("yz_stage=one-launch-lazy" in t) == (mode == "lazy") for t in desc), desc[0]
    scale = max(np.abs(r).max() for r in ref)
    in_scale = float(np.abs(x).max()) * n0 * n1 * n2
    for flags in all_flags:
        # what the lazy plans of this pipeline must reproduce bit for bit: the serial lazy plans, or the two-launch path where the
        # overlapped plan does not use the stage
        same = ("lazy", all_flags[0]) if one[("lazy", flags)] else ("0", flags)
        for d in range(P):
            cnt = ref[d].size
            two, eager, lazy = (res[(m, flags)][0][d][:cnt] for m in ("0", "eager", "lazy"))
            assert np.array_equal(eager, two), (N, P, flags, d)
            assert np.abs(lazy - two).max() / scale < 1e-14, (N, P, flags, d)
            assert np.array_equal(lazy, res[same][0][d][:cnt]), (N, P, flags, d, same)
            for got in (eager, lazy):
                assert np.abs(got.reshape(ref[d].shape) - ref[d]).max() / scale < 1e-11
            cnt = inputs[d].size
            two, eager, lazy = (res[(m, flags)][1][d][:cnt] for m in ("0", "eager", "lazy"))
            assert np.array_equal(eager, two), (N, P, flags, d, "backward")
            assert np.abs(lazy - two).max() / in_scale < 1e-14, (N, P, flags, d, "backward")
            assert np.array_equal(lazy, res[same][1][d][:cnt]), (N, P, flags, d, "backward", same)


@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize("N,P,variants", [((1024, 32, 256), 1, ("half", "fullearly")), ((1024, 128, 256), 4, ("half", "fullearly")),
                                          ((512, 24, 256), 1, ("early",)), ((512, 64, 256), 2, ("early",))])
def test_x_pass_prefetch_variants_are_bit_identical(gpu, N, P, variants, prec, monkeypatch):
    """Forward X pass through the staged transposing store with enough tiles for several grid-stride iterations per workgroup
    (the prefetch carries data from one iteration to the next): the 1024-point kernel with the whole next tile prefetched (the
    default), with half of it, and with the early wait; the 512-point kernel with the early wait (DFFT_X_VARIANT, measurement
    switches) -- from the padded hand-over buffer (P = 1) and from the row-rotated receive buffer (P > 1).  Bit-identical to
    each other, and within the tolerance of the oracle's numpy.fft.fftn."""
    from distributedfft_amd import api
    n0, n1, n2 = N
    monkeypatch.setenv("DFFT_ROT", "1")
    monkeypatch.setenv("DFFT_PAD", "1")
    x = so.random_input(N, seed=n0 + 31 * P)
    ref = so.fftn_reference(x, P)
    inputs = [x[so.slab_start(n0, P, g):so.slab_start(n0, P, g) + so.slab_size(n0, P, g)] for g in range(P)]
    scale = max(np.abs(r).max() for r in ref)
    monkeypatch.delenv("DFFT_X_VARIANT", raising=False)
    base, _ = _run_plans(gpu, N, P, prec, x, +1, api.PLAN_INPUT_FROM_IN, inputs)
    for d in range(P):
        cnt = ref[d].size
        assert np.abs(base[d][:cnt].reshape(ref[d].shape) - ref[d]).max() / scale < TOL[prec], (N, P, d)
    for var in variants:
        monkeypatch.setenv("DFFT_X_VARIANT", var)
        got, _ = _run_plans(gpu, N, P, prec, x, +1, api.PLAN_INPUT_FROM_IN, inputs)
        for d in range(P):
            cnt = ref[d].size
            assert np.array_equal(got[d][:cnt], base[d][:cnt]), (N, P, var, d)


def test_plan_tune_keeps_results_bit_identical(gpu):
    """dfft_plan_tune (plan-time placement measurement of the hand-over buffer): a plan that owns such a buffer -- planes a
    multiple of 1 MiB apart, slab beyond the 256 MiB Infinity Cache -- probes its candidates with the X-pass kernel alone
    (which leaves garbage in the result buffer) and afterwards produces bit for bit what the un-tuned plan produces, in both
    directions, also after a second tuning round."""
    import torch
    from distributedfft_amd import api
    N = (512, 256, 256)
    n = N[0] * N[1] * N[2]
    g = torch.Generator(device=gpu)
    g.manual_seed(7)
    a = torch.complex(torch.rand(n, generator=g, device=gpu, dtype=torch.float64) * 2 - 1,
                      torch.rand(n, generator=g, device=gpu, dtype=torch.float64) * 2 - 1)
    b0, b1 = torch.zeros_like(a), torch.zeros_like(a)
    p0 = api.Plan(*N, a, b0, None, 0, 1, api.FORWARD, api.PLAN_INPUT_FROM_IN)
    p0.execute(api.EXEC_NO_TIMING)
    p0.sync()
    assert p0.tune_report()["kept"] == -1          # never tuned
    p1 = api.Plan(*N, a, b1, None, 0, 1, api.FORWARD, api.PLAN_INPUT_FROM_IN)
    b1.copy_(a * 3)                       # the caller's `out` is not the plan's to scribble on: tune() puts it back
    p1.tune()
    assert torch.equal(b1, a * 3)
    rep = p1.tune_report()
    assert 1 <= len(rep["candidates_ms"]) <= 32 and 0 <= rep["kept"] < len(rep["candidates_ms"]) and rep["kept_retimed_ms"] > 0
    assert min(rep["candidates_ms"]) > 0
    p1.execute(api.EXEC_NO_TIMING)
    p1.sync()
    assert torch.equal(b0, b1)
    p1.tune()                             # a second call measures again
    b1.zero_()
    p1.execute()
    assert len(p1.stage_times()) == 4 and torch.equal(b0, b1)
    # backward plans tune the same buffer from the other side (the inverse X pass writes it)
    c0, c1 = torch.zeros_like(a), torch.zeros_like(a)
    q0 = api.Plan(*N, b0, c0, None, 0, 1, api.BACKWARD, api.PLAN_INPUT_FROM_IN)
    q1 = api.Plan(*N, b0, c1, None, 0, 1, api.BACKWARD, api.PLAN_INPUT_FROM_IN)
    q1.tune()
    assert q1.tune_report()["kept"] >= 0
    for q in (q0, q1):
        q.execute(api.EXEC_NO_TIMING)
        q.sync()
    assert torch.equal(c0, c1)
    assert (c0 / n - a).abs().max().item() < 1e-12
    q0.destroy()
    q1.destroy()
    # spot check against the defining sum (the element-wise full-size checks live in test_gpu_fullsize.py)
    k = (3, 5, 7)
    idx = [torch.arange(m, device=gpu, dtype=torch.float64) for m in N]
    ph = [torch.exp(-2j * np.pi * k[d] * idx[d] / N[d]) for d in range(3)]
    direct = (((a.reshape(N) @ ph[2]) @ ph[1]) * ph[0]).sum()
    got = b1[(k[1] * N[2] + k[2]) * N[0] + k[0]]
    assert abs(complex(direct) - complex(got)) / abs(complex(direct)) < 1e-11
    p0.destroy()
    p1.destroy()


# ---- axes beyond the single-pass range: four-step plans (dfft_long.hip; reference templateFFT.cpp:3972-4106) ---------------
@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize("n", [8192, 16384, 6561, 15625, 10000, 65536, 12288])
def test_long_fft1d_rows_and_cols_vs_numpy(gpu, n, prec):
    import torch
    from distributedfft_amd import api
    rng = np.random.default_rng(n)
    x = rng.uniform(-1, 1, (5, n)) + 1j * rng.uniform(-1, 1, (5, n))
    xt = torch.from_numpy(x).to(gpu).to(_torch_dtype(prec))
    ref = np.fft.fft(x)
    assert _rel_err(api.fft1d_rows(xt, api.FORWARD).cpu().numpy(), ref) < TOL[prec]
    assert _rel_err(api.fft1d_rows(xt, api.BACKWARD).cpu().numpy(), np.fft.ifft(x) * n) < TOL[prec]
    y = xt.clone()
    api.fft1d_rows(y, api.FORWARD, out=y)                      # in place
    assert _rel_err(y.cpu().numpy(), ref) < TOL[prec]
    for width in (24, 7):
        c = rng.uniform(-1, 1, (2, n, width)) + 1j * rng.uniform(-1, 1, (2, n, width))
        ct = torch.from_numpy(c).to(gpu).to(_torch_dtype(prec))
        assert _rel_err(api.fft1d_cols(ct, api.FORWARD).cpu().numpy(), np.fft.fft(c, axis=1)) < TOL[prec]
        assert _rel_err(api.fft1d_cols(ct, api.BACKWARD).cpu().numpy(), np.fft.ifft(c, axis=1) * n) < TOL[prec]


@pytest.mark.parametrize("N,P", [((8192, 16, 24), 1), ((16, 8192, 24), 1), ((24, 16, 8192), 1), ((8192, 32, 16), 4),
                                 ((32, 8192, 16), 2), ((16, 36, 6561), 3), ((10000, 10, 10), 2)])
def test_long_axis_3d_plans_vs_fftn(gpu, N, P):
    """A 3D plan with one axis above 4096 runs the reference's (un-fused) stage structure with four-step transforms on that
    axis: forward against numpy.fft.fftn in the [yy][z][kx] slab layout, then backward / N back to the input."""
    n0, n1, n2 = N
    x = so.random_input(N, seed=sum(N) + P)
    ref = so.fftn_reference(x, P)          # numpy.fft.fftn, split into the devices' [yy][z][kx] slabs
    inputs = [x[so.slab_start(n0, P, g):so.slab_start(n0, P, g) + so.slab_size(n0, P, g)] for g in range(P)]
    outs, _ = _run_plans(gpu, N, P, "f64", x, +1, 0, inputs)
    scale = max(np.abs(r).max() for r in ref)
    for d in range(P):
        got = outs[d][:ref[d].size].reshape(ref[d].shape)
        assert np.abs(got - ref[d]).max() / scale < 1e-11, (N, P, d)
    bouts, _ = _run_plans(gpu, N, P, "f64", None, -1, 0, [r for r in ref])
    for g in range(P):
        got = bouts[g][:inputs[g].size].reshape(inputs[g].shape) / float(n0 * n1 * n2)
        assert np.abs(got - inputs[g]).max() < 1e-10, (N, P, g)


def test_unsupported_length_fails_loudly(gpu):
    import torch
    from distributedfft_amd import api
    a = torch.zeros(11 * 8 * 8, dtype=torch.complex128, device=gpu)
    with pytest.raises(api.DfftError):
        api.Plan(11, 8, 8, a, torch.zeros_like(a), None, 0, 1, api.FORWARD)


@pytest.mark.parametrize("yparts", [1, 2, 4])
@pytest.mark.parametrize("N,P,parts", [((64, 64, 64), 4, 4), ((64, 48, 24), 2, 3), ((25, 10, 16), 4, 2), ((128, 128, 32), 8, 4),
                                       ((100, 64, 12), 4, 5)])
def test_overlap_mode_part_exchange_vs_oracle(gpu, N, P, parts, yparts, monkeypatch):
    """DFFT_PLAN_OVERLAP: t2 cut into X-plane parts behind the chunked Z+Y passes, and (yparts > 1, even Y split) the last
    part cut again into Y sub-blocks so that the X pass of sub-block k overlaps the exchange of sub-block k+1 (t2/t3
    overlap).  With virtual devices the pieces move through the in-process exchange with the same offsets the RCCL path
    uses (uneven slabs included; those fall back to one Y block)."""
    from distributedfft_amd import api
    monkeypatch.setenv("DFFT_OVERLAP_PARTS", str(parts))
    monkeypatch.setenv("DFFT_OVERLAP_YPARTS", str(yparts))
    n0, n1, n2 = N
    x = so.random_input(N, seed=99 + P)
    ref = so.fftn_reference(x, P)
    inputs = [x[so.slab_start(n0, P, g):so.slab_start(n0, P, g) + so.slab_size(n0, P, g)] for g in range(P)]
    outs, times = _run_plans(gpu, N, P, "f64", x, +1, api.PLAN_OVERLAP, inputs)
    scale = max(np.abs(r).max() for r in ref)
    for d in range(P):
        got = outs[d][:ref[d].size].reshape(ref[d].shape)
        assert np.abs(got - ref[d]).max() / scale < 1e-11, f"overlap N={N} P={P} dev={d}"
    outs2, _ = _run_plans(gpu, N, P, "f64", x, +1, api.PLAN_OVERLAP | api.PLAN_INPUT_FROM_IN, inputs)
    for d in range(P):
        assert np.array_equal(outs2[d][:ref[d].size], outs[d][:ref[d].size])
    # backward: the mirror pipeline (inverse X pass per Y sub-block || exchange, Y+Z per X-plane part as it lands) must
    # reproduce the serial backward transform bit for bit and return N x
    serial, _ = _run_plans(gpu, N, P, "f64", None, -1, api.PLAN_INPUT_FROM_IN, [r for r in ref])
    over, _ = _run_plans(gpu, N, P, "f64", None, -1, api.PLAN_OVERLAP | api.PLAN_INPUT_FROM_IN, [r for r in ref])
    for g in range(P):
        cnt = inputs[g].size
        assert np.array_equal(over[g][:cnt], serial[g][:cnt]), f"backward overlap N={N} P={P} dev={g}"
        assert np.abs(over[g][:cnt].reshape(inputs[g].shape) / float(n0 * n1 * n2) - inputs[g]).max() < 1e-10


# shapes whose exchange buffers can carry rotated rows (even splits, power-of-two N2 of at least two cache lines), chosen to
# reach every kernel with a rotated side: 512-point tiles (staged X pass), 16 points per thread (1024), radix-3 Y axis (768),
# the 2048-point DIF-split Y pass and paired-tile X pass, half-line fallbacks, fp32 column pairs and scalar fp32 columns
ROT_SHAPES = [((64, 64, 64), 2), ((64, 64, 64), 4), ((128, 128, 32), 8), ((512, 8, 32), 2), ((1024, 8, 64), 4), ((16, 768, 16), 2),
              ((2048, 8, 16), 2), ((2048, 8, 8), 4), ((8, 2048, 16), 2), ((16, 2048, 32), 4), ((32, 32, 16), 2), ((24, 40, 256), 4),
              # round 6: rotated sides on the half-line tiles of the lean lengths (staged X pass, Y pass storing rotated tiles), 729 points
              ((1536, 8, 16), 2), ((1280, 8, 32), 4), ((1000, 8, 16), 2), ((8, 1536, 16), 2), ((16, 1280, 16), 4), ((8, 1000, 32), 2),
              ((729, 9, 16), 3), ((8, 729, 16), 2)]


@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize("N,P", ROT_SHAPES)
def test_rotated_exchange_rows_vs_oracle(gpu, N, P, prec, monkeypatch):
    """DFFT_ROT=1: the rows of the packed send buffer / the received slab are rotated by three cache lines per X plane (the
    Y pass rotates whole tiles, the X pass every point by its plane) -- the channel-conflict remedy of the P > 1 pipeline,
    switched on by itself only where received planes are a multiple of 256 KiB apart.  Results must equal the un-rotated
    pipeline's bit for bit (serial and overlapped, forward and backward) and the oracle's within the tolerance."""
    from distributedfft_amd import api
    n0, n1, n2 = N
    x = so.random_input(N, seed=4242 + P + n0)
    ref = so.fftn_reference(x, P)
    inputs = [x[so.slab_start(n0, P, g):so.slab_start(n0, P, g) + so.slab_size(n0, P, g)] for g in range(P)]
    scale = max(np.abs(r).max() for r in ref)
    monkeypatch.setenv("DFFT_ROT", "0")
    plain, _ = _run_plans(gpu, N, P, prec, x, +1, api.PLAN_INPUT_FROM_IN, inputs)
    bplain, _ = _run_plans(gpu, N, P, prec, None, -1, api.PLAN_INPUT_FROM_IN, [r for r in ref])
    monkeypatch.setenv("DFFT_ROT", "1")
    for flags in (api.PLAN_INPUT_FROM_IN, api.PLAN_INPUT_FROM_IN | api.PLAN_OVERLAP, 0):
        rot, _ = _run_plans(gpu, N, P, prec, x, +1, flags, inputs)
        for d in range(P):
            cnt = ref[d].size
            assert np.array_equal(rot[d][:cnt], plain[d][:cnt]), f"forward N={N} P={P} flags={flags} dev={d}"
            assert np.abs(rot[d][:cnt].reshape(ref[d].shape) - ref[d]).max() / scale < TOL[prec]
        brot, _ = _run_plans(gpu, N, P, prec, None, -1, flags, [r for r in ref])
        for g in range(P):
            cnt = inputs[g].size
            assert np.array_equal(brot[g][:cnt], bplain[g][:cnt]), f"backward N={N} P={P} flags={flags} dev={g}"


@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize("N,P", [((2048, 8, 32), 1), ((2048, 8, 32), 2), ((8, 2048, 32), 1), ((16, 2048, 32), 2), ((1024, 8, 32), 1),
                                 ((1024, 16, 32), 2), ((512, 16, 32), 1), ((256, 16, 64), 2)])
def test_2048_point_tiles_for_every_tile_count(gpu, N, P, prec, monkeypatch):
    """The 2048-point kernels (paired half-line tiles of the transposing X pass, DIF-split full-line tiles of the other column
    passes) are persistent: a workgroup's first, later and last tiles take different paths through the loop (more so in the
    software-pipelined builds, -DDFFT_DUAL_PIPELINE=1 / -DDFFT_DIF2_PIPELINE=1, whose register layout alternates from tile to tile).
    The same holds for the staged transposed LOAD of the inverse X pass (fft_tload_tiles_kernel: fp32 column pairs of 256 ... 2048
    points, fp64 up to 256 points), which prefetches the next tile's raw elements -- the shapes with a 256- / 512- / 1024-point X axis.
    DFFT_X_GRID / DFFT_Y_GRID cap the persistent grid: 1, 2, 3 and 5 workgroups give every workgroup several tiles, odd and even
    counts, unequal shares.  Results must not depend on the grid (bit for bit, both directions, natural and packed / rotated maps)
    and must match the oracle."""
    from distributedfft_amd import api
    n0, n1, n2 = N
    x = so.random_input(N, seed=2048 + P)
    ref = so.fftn_reference(x, P)
    inputs = [x[so.slab_start(n0, P, g):so.slab_start(n0, P, g) + so.slab_size(n0, P, g)] for g in range(P)]
    scale = max(np.abs(r).max() for r in ref)
    monkeypatch.setenv("DFFT_ROT", "1")
    monkeypatch.delenv("DFFT_X_GRID", raising=False)
    monkeypatch.delenv("DFFT_Y_GRID", raising=False)
    base, _ = _run_plans(gpu, N, P, prec, x, +1, api.PLAN_INPUT_FROM_IN, inputs)
    bbase, _ = _run_plans(gpu, N, P, prec, None, -1, api.PLAN_INPUT_FROM_IN, [r for r in ref])
    for d in range(P):
        cnt = ref[d].size
        assert np.abs(base[d][:cnt].reshape(ref[d].shape) - ref[d]).max() / scale < TOL[prec], (N, P, d)
    for grid in ("1", "2", "3", "5"):
        monkeypatch.setenv("DFFT_X_GRID", grid)
        monkeypatch.setenv("DFFT_Y_GRID", grid)
        got, _ = _run_plans(gpu, N, P, prec, x, +1, api.PLAN_INPUT_FROM_IN, inputs)
        bgot, _ = _run_plans(gpu, N, P, prec, None, -1, api.PLAN_INPUT_FROM_IN, [r for r in ref])
        for d in range(P):
            assert np.array_equal(got[d][:ref[d].size], base[d][:ref[d].size]), (N, P, prec, grid, d)
            assert np.array_equal(bgot[d][:inputs[d].size], bbase[d][:inputs[d].size]), (N, P, prec, grid, d, "backward")


def test_in_place_plans_and_reload(gpu):
    """out == None / out == in selects the in-place mode (bufferDev2 = in, fft_mpi_3d_api.cpp:68-71); input is captured
    at plan time and can be replaced through bufferDev1 (fftSpeed3d_c2c.cpp:78) for repeated executes."""
    import torch
    from distributedfft_amd import api
    N = (32, 64, 16)
    x = so.random_input(N, seed=21)
    ref = so.fftn_reference(x, 1)[0]
    for out_mode in ("none", "same"):
        a = torch.from_numpy(x.reshape(-1).copy()).to(gpu)
        plan = api.Plan(*N, a, None if out_mode == "none" else a, None, 0, 1, api.FORWARD)
        plan.execute()
        plan.sync()
        got = a.cpu().numpy().reshape(ref.shape)
        assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-11
        # a now holds the spectrum; reload a different input through bufferDev1 and execute again
        x2 = so.random_input(N, seed=22)
        plan.load_input(torch.from_numpy(x2.reshape(-1)).to(gpu))
        plan.execute()
        plan.sync()
        ref2 = so.fftn_reference(x2, 1)[0]
        assert np.abs(a.cpu().numpy().reshape(ref2.shape) - ref2).max() / np.abs(ref2).max() < 1e-11
        plan.destroy()
        # in-place backward returns N * x2
        b = a.clone()
        planb = api.Plan(*N, b, None, None, 0, 1, api.BACKWARD)
        planb.execute()
        planb.sync()
        assert np.abs(b.cpu().numpy().reshape(N) / np.prod(N) - x2).max() < 1e-12
        planb.destroy()


def test_degenerate_and_bad_arguments(gpu, native_lib):
    import ctypes as C
    import torch
    from distributedfft_amd import api
    x = torch.zeros(64, dtype=torch.complex128, device=gpu)
    # zero batch is a no-op, not an error
    assert native_lib.dfft_fft1d_rows(x.data_ptr(), x.data_ptr(), 64, 0, 0, 1, None) == 0
    assert native_lib.dfft_fft1d_cols(x.data_ptr(), x.data_ptr(), 8, 8, 0, 0, 1, None) == 0
    # non-positive sizes, bad direction / dtype, missing communicator, wrong communicator size
    h = C.c_void_p()
    for args in [(0, 8, 8, 0, 1), (8, -1, 8, 0, 1), (8, 8, 8, 0, 0), (8, 8, 8, 7, 1)]:
        assert native_lib.dfft_plan_create(C.byref(h), args[0], args[1], args[2], args[3], args[4], x.data_ptr(), None, None, 0, 1, 0) == -1
    assert native_lib.dfft_plan_create(C.byref(h), 8, 8, 8, 0, 1, x.data_ptr(), None, None, 0, 2, 0) == -1
    comm = api.Comm.local(4)
    assert native_lib.dfft_plan_create(C.byref(h), 8, 8, 8, 0, 1, x.data_ptr(), None, comm.handle, 0, 2, 0) == -1
    comm.destroy()
    # DFFT_PLAN_INPUT_FROM_IN needs an out-of-place plan
    big = torch.zeros(512, dtype=torch.complex128, device=gpu)
    assert native_lib.dfft_plan_create(C.byref(h), 8, 8, 8, 0, 1, big.data_ptr(), None, None, 0, 1, api.PLAN_INPUT_FROM_IN) == -1
    # buffers smaller than getMaxDataCount are rejected by the host mirror before reaching the library
    with pytest.raises(ValueError):
        api.Plan(8, 8, 8, x, None, None, 0, 1, api.FORWARD)


@pytest.mark.parametrize("N,P", [((32, 24, 16), 1), ((64, 64, 64), 4), ((25, 10, 16), 4), ((24, 10, 12), 4), ((128, 96, 64), 2)])
@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_natural_order_plans_vs_fftn(gpu, N, P, prec):
    """DFFT_PLAN_NATURAL: X-slabbed natural layout in and out, both directions: forward equals numpy.fftn
    of the whole array cut into X slabs; backward of that returns N * input."""
    from distributedfft_amd import api
    n0, n1, n2 = N
    x = so.random_input(N, seed=5 + n0)
    F = np.fft.fftn(x)
    cut = lambda a: [a[so.slab_start(n0, P, g):so.slab_start(n0, P, g) + so.slab_size(n0, P, g)] for g in range(P)]
    outs, _ = _run_plans(gpu, N, P, prec, None, +1, api.PLAN_NATURAL, cut(x))
    scale = np.abs(F).max()
    got = []
    for g, ref in enumerate(cut(F)):
        o = outs[g][:ref.size].reshape(ref.shape)
        assert np.abs(o - ref).max() / scale < TOL[prec], f"natural forward N={N} P={P} dev={g}"
        got.append(ref)
    backs, _ = _run_plans(gpu, N, P, prec, None, -1, api.PLAN_NATURAL | api.PLAN_INPUT_FROM_IN, got)
    for g, ref in enumerate(cut(x)):
        b = backs[g][:ref.size].reshape(ref.shape) / float(n0 * n1 * n2)
        assert np.abs(b - ref).max() < TOL[prec] * 10, f"natural backward N={N} P={P} dev={g}"


_RCCL_SELF_SCRIPT = r"""
import numpy as np, torch
from distributedfft_amd import api
N = (64, 48, 32)
rng = np.random.default_rng(5)
x = (rng.standard_normal(N) + 1j * rng.standard_normal(N))
ref = np.fft.fftn(x)
dev = torch.device("cuda:0")
comm = api.Comm.rccl(api.Comm.rccl_unique_id(), 1, 0)
def run(data, direction, flags):
    a = torch.from_numpy(np.ascontiguousarray(data).reshape(-1)).to(dev)
    b = torch.zeros_like(a)
    torch.cuda.synchronize()
    p = api.Plan(*N, a, b, comm, 0, 1, direction, flags)
    p.execute(api.EXEC_SYNC_STAGES)       # host-timed stages (the driver's mode), then the asynchronous mode
    p.sync()
    ts = p.stage_times()
    assert all(v >= 0 for v in ts)
    p.execute()
    p.sync()
    t = p.stage_times()
    out = b.cpu().numpy()
    p.destroy()
    return out, t
scale = np.abs(ref).max()
tr = ref.transpose(1, 2, 0)                      # [y][z][kx], the pipeline's output layout
for flags in (api.PLAN_INPUT_FROM_IN, api.PLAN_INPUT_FROM_IN | api.PLAN_OVERLAP, api.PLAN_INPUT_FROM_IN | api.PLAN_UNFUSED):
    out, t = run(x, api.FORWARD, flags)
    assert np.abs(out.reshape(tr.shape) - tr).max() / scale < 1e-11, flags
    assert t[2] > 0 or flags & api.PLAN_OVERLAP, (flags, t)   # the exchange stage really ran
for flags in (api.PLAN_INPUT_FROM_IN, api.PLAN_INPUT_FROM_IN | api.PLAN_OVERLAP):
    out, _ = run(tr, api.BACKWARD, flags)
    assert np.abs(out.reshape(N) / x.size - x).max() < 1e-11, flags
out, _ = run(x, api.FORWARD, api.PLAN_INPUT_FROM_IN | api.PLAN_NATURAL)
assert np.abs(out.reshape(N) - ref).max() / scale < 1e-11
out, _ = run(ref, api.BACKWARD, api.PLAN_INPUT_FROM_IN | api.PLAN_NATURAL)
assert np.abs(out.reshape(N) / x.size - x).max() < 1e-11
comm.destroy()
print("RCCL-SELF-OK")
"""


def test_rccl_grouped_send_recv_on_one_gpu(gpu, tmp_path):
    """The only RCCL traffic one GPU allows: a world-size-1 communicator whose self chunk goes through the grouped
    ncclSend/ncclRecv code of exchange_rccl (DFFT_RCCL_SELF_SENDRECV=1) with the exchange stage forced on at P = 1
    (DFFT_FORCE_EXCHANGE=1) -- forward (serial, overlapped parts on the second stream, unfused), backward and both
    natural-order directions.  Separate process: both switches are read once."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DFFT_RCCL_SELF_SENDRECV="1", DFFT_FORCE_EXCHANGE="1", DFFT_OVERLAP_PARTS="4",
               PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", _RCCL_SELF_SCRIPT], capture_output=True, text=True, timeout=600, env=env,
                       cwd=str(tmp_path))
    assert r.returncode == 0 and "RCCL-SELF-OK" in r.stdout, (r.stdout + r.stderr)[-3000:]


@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize("flags_name", ["fused", "unfused", "natural"])
@pytest.mark.parametrize("N,P", [((32, 48, 24), 1), ((64, 64, 64), 4), ((25, 10, 16), 4)])
def test_plan_scale_is_folded_into_the_transform(gpu, N, P, prec, flags_name):
    """dfft_plan_set_scale: forward * 1/N (heFFTe's scale::full) and backward * 1/N (a normalised inverse) without a
    separate scaling pass -- every pipeline variant applies it exactly once."""
    import torch
    from distributedfft_amd import api
    n0, n1, n2 = N
    flags = {"fused": 0, "unfused": api.PLAN_UNFUSED, "natural": api.PLAN_NATURAL}[flags_name]
    x = so.random_input(N, seed=4242)
    inv = 1.0 / (n0 * n1 * n2)
    comm = api.Comm.local(P) if P > 1 else None
    tdt = _torch_dtype(prec)
    full = np.fft.fftn(x)
    for direction in (api.FORWARD, api.BACKWARD):
        if flags_name == "natural":
            src_full = x if direction == api.FORWARD else full
            want_full = full * inv if direction == api.FORWARD else x          # ifftn(full) = x
            srcs = [src_full[so.slab_start(n0, P, g):so.slab_start(n0, P, g) + so.slab_size(n0, P, g)] for g in range(P)]
            wants = [want_full[so.slab_start(n0, P, g):so.slab_start(n0, P, g) + so.slab_size(n0, P, g)] for g in range(P)]
        elif direction == api.FORWARD:
            srcs = [x[so.slab_start(n0, P, g):so.slab_start(n0, P, g) + so.slab_size(n0, P, g)] for g in range(P)]
            wants = [r * inv for r in so.fftn_reference(x, P)]
        else:
            srcs = so.fftn_reference(x, P)
            wants = [x[so.slab_start(n0, P, g):so.slab_start(n0, P, g) + so.slab_size(n0, P, g)] for g in range(P)]
        plans, outs = [], []
        for g in range(P):
            mc = api.get_max_data_count(n0, n1, n2, P, g == P - 1)
            a = torch.zeros(mc, dtype=tdt, device=gpu)
            b = torch.zeros(mc, dtype=tdt, device=gpu)
            s = torch.from_numpy(np.ascontiguousarray(srcs[g]).reshape(-1)).to(gpu).to(tdt)
            a[:s.numel()] = s
            torch.cuda.synchronize()
            p = api.Plan(n0, n1, n2, a, b, comm, g, P, direction, flags)
            p.set_scale(inv)
            plans.append(p)
            outs.append(b)
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
        scale = max(np.abs(w).max() for w in wants)
        for g in range(P):
            got = outs[g].cpu().numpy()[:wants[g].size].reshape(wants[g].shape)
            assert np.abs(got - wants[g]).max() / scale < TOL[prec] * 10, f"{flags_name} dir={direction} dev={g}"
        for p in plans:
            p.destroy()
    if comm:
        comm.destroy()
