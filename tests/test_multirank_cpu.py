"""CPU multi-process tests of the N > 1 path (world size 2/3, gloo / TCP):
  * the exchange layout the library hands to RCCL (dfft_exchange_layout) drives a gloo all_to_all_single over
    oracle-packed slabs and must reproduce the reference's re-slabbed layout [x][yl][N2] on every rank;
  * the TCP rendezvous (dfft_boot_*) that replaces MPI's control plane: bcast / barrier / allreduce-max."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


GLOO_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["DFFT_ROOT"])
from oracle import slab_oracle as so
from distributedfft_amd import api
N = tuple(int(v) for v in os.environ["DFFT_N"].split("x"))
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world)
n0, n1, n2 = N
x = so.random_input(N, seed=77)
xs0, xsz = so.slab_start(n0, world, rank), so.slab_size(n0, world, rank)
# local stages t0 + t1 by the oracle (checker), counts/offsets from the product's C-ABI
send = so.t1_pack(so.t0_fft_yz(x[xs0:xs0 + xsz], +1), world)
lay = api.exchange_layout(n0, n1, n2, world, rank, api.FORWARD)
assert lay.soffset == sorted(lay.soffset) and lay.roffset == sorted(lay.roffset)
sbuf = torch.from_numpy(np.concatenate([send[o:o + c] for o, c in zip(lay.soffset, lay.scount)]).view(np.float64).copy())
rbuf = torch.empty(2 * sum(lay.rcount), dtype=torch.float64)
dist.all_to_all_single(rbuf, sbuf, [2 * c for c in lay.rcount], [2 * c for c in lay.scount])
recv = np.zeros(api.get_max_data_count(n0, n1, n2, world, rank == world - 1), dtype=np.complex128)
flat = rbuf.numpy().view(np.complex128)
pos = 0
for o, c in zip(lay.roffset, lay.rcount):
    recv[o:o + c] = flat[pos:pos + c]; pos += c
ys = so.slab_size(n1, world, rank)
out = so.t3_fft_x(recv, n0, ys, n2, +1)
ref = so.fftn_reference(x, world)[rank]
err = float(np.abs(out - ref).max() / np.abs(ref).max())
# the re-slabbed buffer is exactly [x][ys][N2] of the YZ-transformed array
yz = np.fft.fft2(x, axes=(1, 2))[:, so.slab_start(n1, world, rank):so.slab_start(n1, world, rank) + ys, :]
lay_err = float(np.abs(recv[:n0 * ys * n2].reshape(n0, ys, n2) - yz).max())
t = torch.tensor([err, lay_err]); dist.all_reduce(t, op=dist.ReduceOp.MAX)
if rank == 0: print("RESULT", t[0].item(), t[1].item())
dist.destroy_process_group()
'''


@pytest.mark.parametrize("N,world", [((16, 12, 8), 2), ((10, 10, 4), 2), ((12, 9, 6), 3)])
def test_gloo_alltoall_with_library_layout(native_lib, N, world):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DFFT_ROOT=str(ROOT), DFFT_N="x".join(map(str, N)))
        procs.append(subprocess.Popen([sys.executable, "-c", GLOO_WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    line = [l for l in outs[0][0].splitlines() if l.startswith("RESULT")][0].split()
    assert float(line[1]) < 1e-12 and float(line[2]) < 1e-11


OVERLAP_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["DFFT_ROOT"])
from oracle import slab_oracle as so
from distributedfft_amd import api
N = tuple(int(v) for v in os.environ["DFFT_N"].split("x"))
PP, YK = int(os.environ["DFFT_PART_PLANES"]), int(os.environ["DFFT_YCUTS"])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world)
n0, n1, n2 = N
x = so.random_input(N, seed=78)
xs0, xs = so.slab_start(n0, world, rank), so.slab_size(n0, world, rank)
yl = so.slab_size(n1, world, 0)
yz = so.t0_fft_yz(x[xs0:xs0 + xs], +1).reshape(xs, n1, n2)           # checker: local YZ stage
if YK > 1:   # send layout [k][dst][x][y in k][N2]
    ysub = yl // YK
    send = np.concatenate([yz[:, d * yl + k * ysub:d * yl + (k + 1) * ysub, :].reshape(-1) for k in range(YK) for d in range(world)])
else:        # send layout [dst][x][y_dst][N2] at the reference's block offsets (oracle pack)
    send = so.t1_pack(so.t0_fft_yz(x[xs0:xs0 + xs], +1), world)
recv = np.full(api.get_max_data_count(n0, n1, n2, world, rank == world - 1), np.nan + 0j, dtype=np.complex128)
nparts = -(-so.slab_size(n0, world, 0) // PP)
moved = 0
for part in range(nparts):
    pieces = [-1] if (part + 1 < nparts or YK == 1) else list(range(YK))   # what execute_forward issues
    for ycut in pieces:
        msgs = api.exchange_part_layout(n0, n1, n2, world, rank, PP, part, YK, ycut)
        ops, bufs = [], []
        for peer, so_, sc, ro, rc in msgs:
            if peer == rank:
                assert sc == rc
                recv[ro:ro + rc] = send[so_:so_ + sc]; moved += sc
                continue
            if sc:
                t = torch.from_numpy(send[so_:so_ + sc].view(np.float64).copy()); ops.append(dist.P2POp(dist.isend, t, peer)); moved += sc
            if rc:
                r = torch.empty(2 * rc, dtype=torch.float64); ops.append(dist.P2POp(dist.irecv, r, peer)); bufs.append((ro, rc, r))
        if ops:
            for w in dist.batch_isend_irecv(ops): w.wait()
        for ro, rc, r in bufs:
            recv[ro:ro + rc] = r.numpy().view(np.complex128)
ys = so.slab_size(n1, world, rank)
assert moved == xs * n1 * n2, (moved, xs * n1 * n2)                 # the pieces together move every element exactly once
got = recv[:n0 * ys * n2]
assert not np.isnan(got).any()
ref = np.fft.fft2(x, axes=(1, 2))[:, so.slab_start(n1, world, rank):so.slab_start(n1, world, rank) + ys, :]   # [x][ys][N2]
if YK > 1:
    ysub = ys // YK
    want = np.concatenate([ref[:, k * ysub:(k + 1) * ysub, :].reshape(-1) for k in range(YK)])   # [k][x][y in k][N2]
else:
    want = ref.reshape(-1)
err = float(np.abs(got - want).max())

# ---- and back: the mirror pieces (Y sub-blocks whole, the last one X-plane part by part) return every rank's X slab ----
bsend = got.copy()                                                   # [k][x all][y in k][N2] ([x][ys][N2] for YK = 1)
brecv = np.full(recv.size, np.nan + 0j, dtype=np.complex128)
blk = so.slab_size(n0, world, 0)
bmoved = 0
for y in range(YK):
    last = (y + 1 == YK)
    for part in (range(nparts) if last else [0]):
        msgs = api.exchange_part_layout(n0, n1, n2, world, rank, PP if last else blk, part, YK, y if YK > 1 else -1, api.BACKWARD)
        ops, bufs = [], []
        for peer, so_, sc, ro, rc in msgs:
            if peer == rank:
                assert sc == rc
                brecv[ro:ro + rc] = bsend[so_:so_ + sc]; bmoved += sc
                continue
            if sc:
                t = torch.from_numpy(bsend[so_:so_ + sc].view(np.float64).copy()); ops.append(dist.P2POp(dist.isend, t, peer)); bmoved += sc
            if rc:
                r = torch.empty(2 * rc, dtype=torch.float64); ops.append(dist.P2POp(dist.irecv, r, peer)); bufs.append((ro, rc, r))
        if ops:
            for w in dist.batch_isend_irecv(ops): w.wait()
        for ro, rc, r in bufs:
            brecv[ro:ro + rc] = r.numpy().view(np.complex128)
assert bmoved == n0 * ys * n2, (bmoved, n0 * ys * n2)
# what arrived is this rank's planes of the YZ-transformed array in the packed layout the forward pipeline sent
if YK > 1:
    back = brecv[:send.size]
    berr = float(np.abs(back - send).max())
else:
    lay = api.exchange_layout(n0, n1, n2, world, rank, api.FORWARD)
    berr = max(float(np.abs(brecv[o:o + c] - send[o:o + c]).max()) for o, c in zip(lay.soffset, lay.scount) if c)
t = torch.tensor([err, berr]); dist.all_reduce(t, op=dist.ReduceOp.MAX)
if rank == 0: print("RESULT", t[0].item(), t[1].item())
dist.destroy_process_group()
'''


@pytest.mark.parametrize("N,world,part_planes,ycuts", [((16, 12, 8), 2, 3, 2), ((16, 12, 8), 2, 8, 1), ((10, 10, 4), 2, 2, 1),
                                                       ((24, 24, 6), 4, 2, 3), ((12, 9, 6), 3, 1, 1)])
def test_gloo_overlapped_exchange_pieces(native_lib, N, world, part_planes, ycuts):
    """The piece-wise exchange of DFFT_PLAN_OVERLAP (X-plane parts, last part per Y sub-block) driven over gloo send/recv
    with the message lists the library issues to RCCL (dfft_exchange_part_layout): every element moves exactly once and the
    receive buffer ends up as [k][x][y in k][N2] of the YZ-transformed array on every rank (uneven slabs for ycuts = 1)."""
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DFFT_ROOT=str(ROOT), DFFT_N="x".join(map(str, N)), DFFT_PART_PLANES=str(part_planes), DFFT_YCUTS=str(ycuts))
        procs.append(subprocess.Popen([sys.executable, "-c", OVERLAP_WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2500:]
    line = [l for l in outs[0][0].splitlines() if l.startswith("RESULT")][0].split()
    assert float(line[1]) < 1e-11 and float(line[2]) == 0.0   # forward pieces; backward pieces return the sent data exactly


BOOT_WORKER = r'''
import os, sys, ctypes as C
sys.path.insert(0, os.environ["DFFT_ROOT"])
from distributedfft_amd import _lib
lib = _lib.load()
assert lib.dfft_boot_init() == 0, lib.dfft_last_error()
r, n = lib.dfft_boot_rank(), lib.dfft_boot_size()
buf = C.create_string_buffer(128)
if r == 0: buf.raw = bytes(range(128))
assert lib.dfft_boot_bcast(buf, 128, 0) == 0
assert buf.raw == bytes(range(128))
v = (C.c_double * 2)(float(r), -float(r))
assert lib.dfft_boot_allreduce_max(v, 2) == 0
assert (v[0], v[1]) == (float(n - 1), 0.0), (v[0], v[1])
for _ in range(3): assert lib.dfft_boot_barrier() == 0
assert lib.dfft_boot_finalize() == 0
print("BOOT_OK", r, n)
'''


@pytest.mark.parametrize("world", [1, 2, 4])
def test_tcp_rendezvous(native_lib, world):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, DFFT_RANK=str(r), DFFT_WORLD_SIZE=str(world), DFFT_MASTER_ADDR="127.0.0.1",
                   DFFT_MASTER_PORT=str(port), DFFT_ROOT=str(ROOT))
        for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        procs.append(subprocess.Popen([sys.executable, "-c", BOOT_WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    for r, p in enumerate(procs):
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0, e[-2000:]
        assert f"BOOT_OK {r} {world}" in o


DEAD_PEER_WORKER = r'''
import os, sys, time
sys.path.insert(0, os.environ["DFFT_ROOT"])
from distributedfft_amd import _lib
lib = _lib.load()
assert lib.dfft_boot_init() == 0, lib.dfft_last_error()
assert lib.dfft_boot_barrier() == 0
if lib.dfft_boot_rank() == int(os.environ["DFFT_TEST_SLEEPER"]):
    time.sleep(60)                      # leaves the collective call sequence: its peers must not wait for ever
t0 = time.monotonic()
rc = lib.dfft_boot_barrier()
print("BARRIER", rc, round(time.monotonic() - t0, 1), lib.dfft_last_error().decode(), flush=True)
'''


@pytest.mark.parametrize("sleeper", [0, 1])
def test_rendezvous_waits_are_bounded(native_lib, sleeper):
    """A rank that stops taking part (round 4's "stalls" began with a rank that had died): the others' barrier returns DFFT_ECOMM
    after DFFT_BOOT_TIMEOUT_S naming the collective, the rank waited for and the reason, and prints the process's last
    control-plane events (dfft_trace.cpp) -- instead of blocking in recv() for ever."""
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, DFFT_RANK=str(r), DFFT_WORLD_SIZE="2", DFFT_MASTER_ADDR="127.0.0.1", DFFT_MASTER_PORT=str(port),
                   DFFT_ROOT=str(ROOT), DFFT_BOOT_TIMEOUT_S="3", DFFT_TEST_SLEEPER=str(sleeper))
        for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        procs.append(subprocess.Popen([sys.executable, "-c", DEAD_PEER_WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    waiter = procs[1 - sleeper]
    o, e = waiter.communicate(timeout=60)
    procs[sleeper].kill()
    procs[sleeper].communicate()
    line = [l for l in o.splitlines() if l.startswith("BARRIER")][0].split(None, 3)
    assert int(line[1]) == -5 and 2.0 <= float(line[2]) <= 15.0, o          # DFFT_ECOMM, after about the limit
    assert f"waiting for rank {sleeper}" in line[3] and "DFFT_BOOT_TIMEOUT_S" in line[3]
    assert "[dfft trace]" in e and "boot: allreduce/barrier enter" in e


def test_rendezvous_rejects_a_stranger_on_its_port(native_lib):
    """The hello is {magic, rank} and is acknowledged with {magic, world size}: something else that connects to rank 0's port is
    dropped without failing the job, and a rank that finds a foreign listener on the port keeps trying until the real rank 0 is there."""
    import socket
    import time
    port = _free_port()
    foreign = socket.socket()                      # a listener that is NOT rank 0 owns the port first
    foreign.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    foreign.bind(("127.0.0.1", port))
    foreign.listen(4)
    env1 = dict(os.environ, DFFT_RANK="1", DFFT_WORLD_SIZE="2", DFFT_MASTER_ADDR="127.0.0.1", DFFT_MASTER_PORT=str(port), DFFT_ROOT=str(ROOT))
    for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env1.pop(k, None)
    p1 = subprocess.Popen([sys.executable, "-c", BOOT_WORKER], env=env1, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    c, _ = foreign.accept()                        # rank 1 took the stranger for rank 0 ...
    c.close()                                      # ... which does not answer with the magic word
    foreign.close()
    time.sleep(0.3)
    p0 = subprocess.Popen([sys.executable, "-c", BOOT_WORKER], env=dict(env1, DFFT_RANK="0"), stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True)
    time.sleep(0.5)
    try:                                           # and a stranger knocks at the real rank 0's door
        s = socket.create_connection(("127.0.0.1", port), timeout=2)
        s.sendall(b"GET / HTTP/1.0\r\n\r\n")
        s.close()
    except OSError:
        pass
    for r, p in ((0, p0), (1, p1)):
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0, e[-2000:]
        assert f"BOOT_OK {r} 2" in o


@pytest.mark.parametrize("world", [1, 2, 4])
def test_bench_control_plane_dry_run(native_lib, world):
    """bench.py's multi-rank plumbing (torchrun-style env, gloo rendezvous, 128-byte id broadcast, slab bookkeeping),
    without touching a GPU: `--dry-run` stops before any device work."""
    import json
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "bench.py"), "--gpus", str(world), "--dry-run"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    d = json.loads(outs[0][0].strip().splitlines()[-1])
    assert d["ok"] and d["n_gpus"] == world and d["elements_exchanged"] == 512 ** 3 and d["local_count"] == 512 ** 3 // world


STUB_WORKER = r'''
import os, sys, runpy, json
sys.path.insert(0, os.environ["DFFT_ROOT"])
import torch
from distributedfft_amd import api

# Replace the native plan / communicator objects by stubs: NOTHING is computed; this only drives bench.py's multi-rank
# control flow and report assembly (the part that cannot be exercised on a single-GPU box).
class StubPlan:
    def __init__(self, n0, n1, n2, a, b, comm, rank, P, direction, flags=0):
        self.a, self.b, self.direction, self.flags = a, b, direction, flags
    def execute(self, flags=0):
        n = min(self.a.numel(), self.b.numel())
        self.b[:n] = self.a[:n] * (1.0 if self.direction > 0 else 4096.0)   # "backward(forward(x)) = N x" for 16^3
        if (self.flags & api.PLAN_OVERLAP) and os.environ.get("STUB_BREAK_OVERLAP") == os.environ["RANK"]:
            self.b[3] += 1.0   # an overlapped pipeline that corrupts one element on one rank
    def sync(self): pass
    def stage_times(self): return [1e-3, 0.0, (5e-4 if self.flags & api.PLAN_OVERLAP else 2e-3), 1e-3]
    def kernel_times(self): raise api.DfftError(-1, "stub", "interleaved")
    def destroy(self): pass
class StubComm:
    def __init__(self, P, r): self.P, self.r = P, r
    def info(self): return {"kind": "rccl", "size": self.P, "rank": self.r, "device": 0}
    def destroy(self): pass
api._BENCH_STUB = True
api.Plan = StubPlan
api.Comm.rccl_unique_id = staticmethod(lambda: bytes(range(128)))
api.Comm.rccl = staticmethod(lambda uid, P, r: StubComm(P, r))
sys.argv = ["bench.py", "--gpus", os.environ["WORLD_SIZE"], "--size", "16", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
runpy.run_path(os.path.join(os.environ["DFFT_ROOT"], "bench.py"), run_name="__main__")
'''


@pytest.mark.parametrize("world", [2, 4])
def test_bench_multirank_report_with_stubbed_plans(native_lib, world):
    """Every Python statement of bench.py's P > 1 branch (gloo rendezvous, id broadcast, timing reduction, overlap and
    serial diagnostics, JSON assembly) with the native objects stubbed out -- the RCCL data path itself needs >= 2 GPUs."""
    import json
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DFFT_ROOT=str(ROOT))
        procs.append(subprocess.Popen([sys.executable, "-c", STUB_WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    d = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == world and d["scaling"] == "strong" and d["unit"] == "GFlops/s" and d["value"] > 0
    assert d["stages_ms"]["t2"] == 0.5 and d["stages_ms_without_overlap"]["t2"] == 2.0
    assert d["overlap_result_bit_identical"] is True and d["xgmi"]["t2_full_ms"] == 2.0
    assert d["roofline"]["kernel"].startswith("fft_cols X") and d["roundtrip_abs_error"] < 1e-12
    assert ("cpu_baseline" not in d) and d["vs_baseline"] is None
    for o, _ in outs[1:]:
        assert not [l for l in o.splitlines() if l.startswith("{")]  # only rank 0 prints


def test_bench_falls_back_to_serial_pipeline_when_overlap_result_differs(native_lib):
    """The overlapped pipeline is refereed by the serial one before anything is timed: one differing element on one rank
    makes every rank time (and report) the serial pipeline, flagged in the JSON."""
    import json
    world, port = 2, _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DFFT_ROOT=str(ROOT), STUB_BREAK_OVERLAP="1")
        procs.append(subprocess.Popen([sys.executable, "-c", STUB_WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    d = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    assert "overlap_fallback" in d and "serial pipeline" in d["overlap_fallback"]
    assert d["stages_ms"]["t2"] == 2.0 and "stages_ms_without_overlap" not in d   # the serial plan's stage times
    assert "overlapped" not in d["config"]["exchange"] and d["value"] > 0
