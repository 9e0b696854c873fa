"""-m gpu: the one-process-per-device path with REAL data on one GPU.

RCCL refuses two ranks on the same device, so the multi-process data path cannot be exercised with it on a single-GPU box.
The IPC communicator (hipIpc-shared receive buffers, device-to-device pushes, rendezvous barriers) has no such limit: W
processes share cuda:0, each owning one slab, and run exactly what W GPUs would run -- plan creation / destruction as
collectives, serial and overlapped pipelines in both directions, natural-order plans, and bench.py's whole N > 1 flow
(referee, timing reduction, serial diagnostic, round trip, direct-DFT spot check over all ranks)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    """A port p with p AND p + 1 free: launched through MASTER_PORT (torchrun's variable) the library's rendezvous listens on
    MASTER_PORT + 1, off torchrun's own store port (dfft_bootstrap.cpp) -- checking p alone left that one to chance (seen once in round 6:
    'bind/listen on 127.0.0.1:38510: Address already in use')."""
    for _ in range(64):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        p = s.getsockname()[1]
        s2 = socket.socket()
        try:
            s2.bind(("127.0.0.1", p + 1))
        except OSError:
            continue
        finally:
            s2.close()
            s.close()
        return p
    raise RuntimeError("no pair of free ports")


def _launch(world, argv, extra_env=None, timeout=300, expect_rc=0):
    """One attempt, no retry.  Round 4 repeated a failed launch once because two cases had "stalled for minutes and passed when
    repeated"; round 5 found what that was (profiles/r05/README.md section 1): not a stall but a WRONG RESULT on one rank -- a
    second generation of plans on one IPC communicator whose re-exported receive buffer a peer had mapped onto stale memory, or
    whose plan-time input copy ran late on the null stream -- after which the rank's assertion killed it and its peers waited for
    it in the next exchange.  Both causes are fixed in the library (pooled registrations, plan-time copy on the plan's stream), the
    rendezvous' waits are bounded, and a launch that fails is a finding again."""
    return _launch_once(world, argv, extra_env, timeout, expect_rc)


def _launch_once(world, argv, extra_env=None, timeout=300, expect_rc=0):
    """One process per rank.  A rank that fails (or a hang) is a finding, not just a time-out: as soon as one rank has exited with an
    unexpected code the others get a few seconds to notice and are then killed -- peers of a dead rank wait in the next rendezvous
    for ever otherwise -- and the assertion shows what every rank had printed."""
    import tempfile
    import time
    port = _free_port()
    procs, files = [], []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DFFT_ROOT=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   DFFT_BENCH_ALLOW_SHARED_GPU="1")  # these tests put several ranks on one GPU on purpose
        env.pop("DFFT_MASTER_PORT", None)
        if extra_env:
            env.update(extra_env)
        fo, fe = tempfile.TemporaryFile("w+"), tempfile.TemporaryFile("w+")  # files, not pipes: nobody blocks on a full pipe
        files.append((fo, fe))
        procs.append(subprocess.Popen(argv, env=env, stdout=fo, stderr=fe, text=True, cwd=str(ROOT)))
    t_end = time.monotonic() + timeout
    failed_at = None
    why = ""
    while any(p.poll() is None for p in procs):
        now = time.monotonic()
        if failed_at is None and any(p.poll() is not None and p.returncode != expect_rc for p in procs):
            failed_at = now
        if now > t_end or (failed_at is not None and now > failed_at + 10.0):
            why = (f"{world} ranks did not finish within {timeout} s" if now > t_end
                   else "a rank exited with an unexpected code; its peers were killed 10 s later")
            for q in procs:
                if q.poll() is None:
                    q.kill()
            for q in procs:
                q.wait()
            break
        time.sleep(0.1)
    outs = []
    for fo, fe in files:
        fo.seek(0)
        fe.seek(0)
        outs.append((fo.read(), fe.read()))
        fo.close()
        fe.close()
    if why or any(p.returncode != expect_rc for p in procs):
        tails = "\n".join(f"--- rank {r} rc={p.returncode}\n{(o + e)[-2500:]}" for r, (p, (o, e)) in enumerate(zip(procs, outs)))
        raise AssertionError((why or f"unexpected exit code (expected {expect_rc})") + "\n" + tails)
    return outs


def test_bench_refuses_ranks_that_share_a_device(gpu):
    """`bench.py --gpus 2` with both ranks on ONE device exits non-zero on every rank (a number measured that way would carry
    the name of a 2-GPU run); the functional tests below opt out explicitly with DFFT_BENCH_ALLOW_SHARED_GPU=1."""
    outs = _launch(2, [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--size", "64", "--steps", "2", "--warmup", "1",
                       "--no-cpu-baseline"],
                   {"DFFT_EXCHANGE": "ipc-async", "DFFT_BENCH_ALLOW_SHARED_GPU": "0",
                    "HIP_VISIBLE_DEVICES": os.environ.get("HIP_VISIBLE_DEVICES", "0").split(",")[0]}, timeout=300, expect_rc=4)
    assert "distinct device" in outs[0][1] and not any(l.startswith("{") for l in outs[0][0].splitlines())


def _device_count():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_bench_two_gpus_on_rccl(gpu):
    """The north-star exchange on real links: `bench.py --gpus 2` as the driver launches it, RCCL grouped send/recv between
    two devices.  The report must say RCCL ran (no fall-back), the overlapped pipeline must match the serial one bit for bit,
    and the result must pass the direct-DFT spot check over both ranks' inputs."""
    outs = _launch(2, [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--size", "256", "--steps", "5", "--warmup", "2",
                       "--no-cpu-baseline"], {"DFFT_EXCHANGE": "rccl", "DFFT_BENCH_ALLOW_SHARED_GPU": "0"}, timeout=600)
    d = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and "RCCL" in d["config"]["exchange"] and "exchange_fallback" not in d
    assert d["overlap_result_bit_identical"] is True
    assert d["roundtrip_abs_error"] < 1e-11 and d["direct_dft_spot_check_rel_error"] < 1e-11
    for r, (_, e) in enumerate(outs):
        assert "ncclCommInitRank ok: RCCL reports 2 ranks" in e, e[-2000:]


@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_speedtest_two_ranks_on_rccl(gpu):
    """`sh speedTest.sh 2 64 64 64` (the reference's CLI, speedTest.sh:1-17) on the default RCCL exchange: the driver's own
    report block with a sane error."""
    import re
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DFFT_MASTER_PORT=str(_free_port()))
    env.pop("DFFT_EXCHANGE", None)
    r = subprocess.run(["sh", str(ROOT / "speedTest.sh"), "2", "64", "64", "64"], capture_output=True, text=True, env=env,
                       cwd=str(ROOT), timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "Performance:" in r.stdout and "t0:" in r.stdout
    assert float(re.search(r"Max error:\s*([0-9.eE+-]+)", r.stdout).group(1)) < 1e-11


WORKER = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["DFFT_ROOT"])
from distributedfft_amd import api
N = tuple(int(v) for v in os.environ["DFFT_N"].split("x"))
rank, P = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
n0, n1, n2 = N
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
comm = api.Comm.ipc(P, rank, os.environ.get("DFFT_IPC_ASYNC") == "1")
rng = np.random.default_rng(2024)
x = rng.standard_normal(N) + 1j * rng.standard_normal(N)          # same array on every rank
full = np.fft.fftn(x)
xb = -(-n0 // P); x0 = rank * xb; xs = min(xb, n0 - x0)
yb = -(-n1 // P); y0 = rank * yb; ys = min(yb, n1 - y0)
mc = api.get_max_data_count(n0, n1, n2, P, rank == P - 1)
scale = np.abs(full).max()

def run(src, direction, flags, reps=2):
    a = torch.zeros(mc, dtype=torch.complex128, device=dev)
    b = torch.zeros(mc, dtype=torch.complex128, device=dev)
    a[:src.size] = torch.from_numpy(np.ascontiguousarray(src).reshape(-1)).to(dev)
    torch.cuda.synchronize()
    p = api.Plan(n0, n1, n2, a, b, comm, rank, P, direction, flags)   # collective
    for _ in range(reps):
        p.execute(api.EXEC_NO_TIMING)
    p.sync()
    out = b.cpu().numpy()
    p.destroy()                                                        # collective
    return out

mine_in = x[x0:x0 + xs]
want_fwd = full[:, y0:y0 + ys, :].transpose(1, 2, 0)               # [yy][z][kx]
outs = {}
for name, flags in (("serial", api.PLAN_INPUT_FROM_IN), ("overlap", api.PLAN_INPUT_FROM_IN | api.PLAN_OVERLAP),
                    ("unfused", api.PLAN_INPUT_FROM_IN | api.PLAN_UNFUSED), ("inplace-style", api.PLAN_DEFAULT)):
    o = run(mine_in, api.FORWARD, flags, reps=1 if name == "inplace-style" else 2)
    outs[name] = o
    assert np.abs(o[:want_fwd.size].reshape(want_fwd.shape) - want_fwd).max() / scale < 1e-11, name
assert np.array_equal(outs["serial"][:want_fwd.size], outs["overlap"][:want_fwd.size])   # bit-identical pipelines
for name, flags in (("serial", api.PLAN_INPUT_FROM_IN), ("overlap", api.PLAN_INPUT_FROM_IN | api.PLAN_OVERLAP)):
    o = run(want_fwd, api.BACKWARD, flags)
    assert np.abs(o[:mine_in.size].reshape(mine_in.shape) / x.size - mine_in).max() < 1e-11, "backward " + name
o = run(mine_in, api.FORWARD, api.PLAN_INPUT_FROM_IN | api.PLAN_NATURAL)
assert np.abs(o[:mine_in.size].reshape(mine_in.shape) - full[x0:x0 + xs]).max() / scale < 1e-11, "natural forward"
o = run(full[x0:x0 + xs], api.BACKWARD, api.PLAN_INPUT_FROM_IN | api.PLAN_NATURAL)
assert np.abs(o[:mine_in.size].reshape(mine_in.shape) / x.size - mine_in).max() < 1e-11, "natural backward"
comm.destroy()
print("MP-OK", rank)
'''


@pytest.mark.parametrize("async_exchange", ["0", "1"], ids=["host-sync", "stream-ordered"])
@pytest.mark.parametrize("N,world", [((64, 64, 32), 2), ((64, 64, 64), 4), ((25, 10, 16), 4), ((48, 40, 12), 3)])
def test_one_process_per_slab_on_one_gpu(gpu, N, world, async_exchange):
    """async_exchange = 1: the exchange is enqueued on the plan's streams (flag words in IPC-shared memory, published and
    awaited by one-wave kernels), i.e. the overlapped pipelines run with the same asynchrony they have on RCCL -- X passes
    racing later sub-blocks' copies, Y+Z passes racing later parts -- across real process boundaries."""
    outs = _launch(world, [sys.executable, "-c", WORKER], {"DFFT_N": "x".join(map(str, N)), "DFFT_IPC_ASYNC": async_exchange})
    for r, (o, _) in enumerate(outs):
        assert f"MP-OK {r}" in o


@pytest.mark.parametrize("ranks,overlap", [(2, "0"), (4, "1")])
def test_speedtest_sh_multi_process_driver(gpu, tmp_path, ranks, overlap):
    """sh speedTest.sh <ranks> X Y Z with one process per rank (the reference's mpirun -np <ranks> ./distFFTOpt X Y Z 1,
    speedTest.sh:6): TCP rendezvous instead of MPI, hipIpc exchange so the ranks can share the one GPU; the driver's own
    report block and round-trip error metric, also with the overlapped forward pipeline (DFFT_OVERLAP=1)."""
    import re
    env = dict(os.environ, DFFT_EXCHANGE="ipc", DFFT_MASTER_PORT=str(_free_port()), DFFT_OVERLAP=overlap,
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run(["sh", str(ROOT / "speedTest.sh"), str(ranks), "64", "32", "48"], capture_output=True, text=True,
                       timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = r.stdout
    assert re.search(r"Size:\s*64x32x48", out) and re.search(rf"MPI ranks:\s*{ranks}\b", out)
    assert float(re.search(r"Max error:\s*([0-9.eE+-]+)", out).group(1)) < 1e-11
    assert float(re.search(r"Performance:\s*([0-9.eE+-]+)", out).group(1)) > 0
    nlines = out.count("t0: ")              # one stage line per forward execute per rank (the ranks share the pipe)
    assert nlines >= 2 * ranks and nlines % ranks == 0


@pytest.mark.parametrize("world,precision,tol", [(2, "double", 1e-11), (3, "float", 5e-4)])
def test_heffte_protocol_front_end_multi_process(gpu, world, precision, tol):
    """speed3d_c2c (heFFTe's benchmark protocol on the C-ABI) with one process per slab: natural-order plans, i.e. two
    all-to-alls per transform, forward scaled by 1/N in the X pass, backward, heFFTe's tolerance."""
    import re
    from distributedfft_amd import _lib
    exe = _lib.LIB_PATH.parent / "speed3d_c2c"
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, DFFT_RANK=str(r), DFFT_WORLD_SIZE=str(world), DFFT_MASTER_ADDR="127.0.0.1",
                   DFFT_MASTER_PORT=str(port), DFFT_LOCAL_DEVICE="0", DFFT_EXCHANGE="ipc", HSA_ENABLE_IPC_MODE_LEGACY="0")
        for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_RANK"):
            env.pop(k, None)
        procs.append(subprocess.Popen([str(exe), "stock", precision, "48", "36", "24", "-slabs"], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, (o + e)[-2000:]
    out = outs[0][0]
    assert "heFFTe performance test" in out and "Size:      48x36x24" in out
    assert float(re.search(r"Max error:\s*([0-9.eE+-]+)", out).group(1)) < tol
    assert float(re.search(r"Time per run:\s*([0-9.eE+-]+)", out).group(1)) > 0


def test_bench_falls_back_to_ipc_when_rccl_cannot_start(gpu):
    """Default exchange (RCCL) with two ranks on ONE GPU: RCCL cannot serve duplicate devices -- bench.py knows from the PCI
    addresses it gathered and does not call ncclCommInitRank at all (round 5; before, RCCL was driven into its error on purpose,
    which once left a rank inside the call until the 180 s watchdog) -- every rank agrees on the failure and the run continues on
    the stream-ordered IPC communicator; the report says so."""
    outs = _launch(2, [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--size", "64", "--steps", "3", "--warmup", "1",
                       "--no-cpu-baseline"],
                   {"DFFT_EXCHANGE": "rccl", "NCCL_DEBUG": "WARN",
                    # one visible device (the first of whatever is visible now): both ranks land on it even on a multi-GPU box
                    "HIP_VISIBLE_DEVICES": os.environ.get("HIP_VISIBLE_DEVICES", "0").split(",")[0]}, timeout=300)
    d = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    assert "exchange_fallback" in d and "RCCL" in d["exchange_fallback"] and "ipc-async" in d["config"]["exchange"].lower()
    assert d["overlap_result_bit_identical"] is True and d["direct_dft_spot_check_rel_error"] < 1e-11


@pytest.mark.parametrize("backend", ["ipc", "ipc-async"])
@pytest.mark.parametrize("world,size", [(2, "64"), (4, "128")])
def test_bench_multirank_flow_with_real_data(gpu, world, size, backend):
    """bench.py --gpus W exactly as the driver launches it (one rank per process, torchrun-style environment), with the IPC
    exchange so that all ranks can share the one GPU: every check of the N > 1 report on real transforms."""
    outs = _launch(world, [sys.executable, str(ROOT / "bench.py"), "--gpus", str(world), "--size", size, "--steps", "4",
                           "--warmup", "2", "--no-cpu-baseline"], {"DFFT_EXCHANGE": backend})
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and outs[0][0].strip() == lines[0]          # exactly one line on rank 0's stdout
    for o, _ in outs[1:]:
        assert o.strip() == ""                                           # and nothing on the others'
    d = json.loads(lines[0])
    n = int(size)
    assert d["n_gpus"] == world and d["scaling"] == "strong" and d["value"] > 0 and d["dtype"] == "f64"
    assert d["overlap_result_bit_identical"] is True                    # the referee: overlapped == serial, bit for bit
    assert d["roundtrip_abs_error"] < 1e-11 and d["direct_dft_spot_check_rel_error"] < 1e-11
    assert set(d["stages_ms"]) == {"t0", "t1", "t2", "t3"} and set(d["pipeline_probe_ms_per_step"]) == {"overlapped", "serial"}
    if d["pipeline"] == "serial":   # expected with the host-synchronising exchange: there the overlap cannot win
        assert "faster" in d["overlap_fallback"] and d["stages_ms"]["t2"] > 0
    else:
        assert d["stages_ms_without_overlap"]["t2"] > 0
    assert d["xgmi"]["pair_chunk_bytes"] == 16 * n ** 3 // world ** 2 and "ipc" in d["config"]["exchange"].lower()
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["achieved"] > 0


STRESS_WORKER = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["DFFT_ROOT"])
from distributedfft_amd import api
N = tuple(int(v) for v in os.environ["DFFT_N"].split("x"))
rank, P = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
n0, n1, n2 = N
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
comm = api.Comm.ipc(P, rank, True)
mc = api.get_max_data_count(n0, n1, n2, P, rank == P - 1)
g = torch.Generator(device=dev); g.manual_seed(7 + rank)
a = torch.zeros(mc, dtype=torch.complex128, device=dev)
cnt = (n0 // P) * n1 * n2
a[:cnt] = torch.complex(torch.rand(cnt, generator=g, device=dev, dtype=torch.float64), torch.rand(cnt, generator=g, device=dev, dtype=torch.float64))
for direction in (api.FORWARD, api.BACKWARD):
    bs, bo = torch.zeros_like(a), torch.zeros_like(a)
    ps = api.Plan(n0, n1, n2, a, bs, comm, rank, P, direction, api.PLAN_INPUT_FROM_IN)
    po = api.Plan(n0, n1, n2, a, bo, comm, rank, P, direction, api.PLAN_INPUT_FROM_IN | api.PLAN_OVERLAP)
    ps.execute(api.EXEC_NO_TIMING); ps.sync()
    ref = bs.clone()
    for rep in range(3):
        for _ in range(12):                      # a deep queue of overlapped executes, no host synchronisation in between
            po.execute(api.EXEC_NO_TIMING)
        po.sync()
        assert torch.equal(bo, ref), (direction, rep)
    po.destroy(); ps.destroy()
comm.destroy()
print("STRESS-OK", rank)
'''


@pytest.mark.parametrize("N,world,parts,yparts", [((128, 128, 64), 4, "4", "2"), ((128, 64, 128), 2, "8", "4"),
                                                   ((96, 96, 48), 4, "2", "1")])
def test_overlapped_pipeline_stress_across_processes(gpu, N, world, parts, yparts):
    """Back-to-back overlapped executes (12 deep, nothing synchronises the host in between) on the stream-ordered IPC
    exchange, forward and backward, against the serial pipeline's result bit for bit: buffer reuse across consecutive
    executes, send data still in flight when the next pass starts, early X passes -- the races an asynchronous exchange
    can expose and a host-synchronising one cannot."""
    outs = _launch(world, [sys.executable, "-c", STRESS_WORKER],
                   {"DFFT_N": "x".join(map(str, N)), "DFFT_OVERLAP_PARTS": parts, "DFFT_OVERLAP_YPARTS": yparts}, timeout=240)
    for r, (o, _) in enumerate(outs):
        assert f"STRESS-OK {r}" in o


DEAD_PEER_WORKER = r'''
import os, sys, time
import torch
sys.path.insert(0, os.environ["DFFT_ROOT"])
from distributedfft_amd import api
from distributedfft_amd._lib import DfftError
rank, P = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
N = (64, 64, 32)
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
comm = api.Comm.ipc(P, rank, True)
mc = api.get_max_data_count(*N, P, rank == P - 1)
a = torch.ones(mc, dtype=torch.complex128, device=dev)
b = torch.zeros_like(a)
p = api.Plan(*N, a, b, comm, rank, P, api.FORWARD, api.PLAN_INPUT_FROM_IN | api.PLAN_OVERLAP)
p.execute(api.EXEC_NO_TIMING); p.sync()
if rank == 1:
    sys.stdout.flush()
    os._exit(0)                          # dies without a word, plans and communicator alive
time.sleep(1.0)
t0 = time.monotonic()
try:
    for _ in range(12):                  # ~60 exchange rounds queued behind a peer that will never answer
        p.execute(api.EXEC_NO_TIMING)
    p.sync()
    print("NO-ERROR")
except DfftError as e:
    print("DEAD-PEER", round(time.monotonic() - t0, 1), e.code, str(e)[:160], flush=True)
sys.stdout.flush()
os._exit(0)                              # (the collective tear-down would wait for the dead rank as well)
'''


def test_dead_peer_costs_one_time_limit(gpu):
    """A rank that dies mid-job (round 4's "stalls" were peers waiting behind one): the survivor's queued rounds of the stream-ordered
    IPC exchange give up after ONE time limit (DFFT_IPC_TIMEOUT_S) -- the first round that times out makes the later ones return at
    once -- and dfft_plan_sync reports DFFT_ECOMM, instead of one limit per queued round."""
    outs = _launch(2, [sys.executable, "-c", DEAD_PEER_WORKER], {"DFFT_IPC_TIMEOUT_S": "3"}, timeout=120)
    line = [l for l in outs[0][0].splitlines() if l.startswith("DEAD-PEER")]
    assert line, outs[0][0] + outs[0][1][-1500:]
    _, secs, code, _ = line[0].split(None, 3)
    assert int(code) == -5 and 2.0 <= float(secs) <= 20.0, line[0]      # DFFT_ECOMM after about one limit, not 60 of them
