"""-m gpu: heFFTe's own benchmark (benchmarks/speed3d_c2c.cpp, unchanged, built from /root/reference by oracle/Makefile) running
on include/heffte_backend_dfft.h: heFFTe's plan logic, packing and MPI reshapes with every 1-D FFT computed by the gfx950
kernels.  The benchmark validates itself -- forward with 1/N scaling + backward against the input, tolerance 1e-11 (double) /
5e-4 (float), test_common.h:136-140 -- and prints the report block only when that holds (speed3d.h:139-150)."""
import os
import re
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
EXE = ROOT / "oracle" / "_ref" / "speed3d_c2c_dfft"
MPIRUN = Path("/opt/conda/bin/mpirun")


def _run(np_, args):
    if not EXE.exists() or not MPIRUN.exists():
        pytest.skip("oracle/_ref/speed3d_c2c_dfft not built (needs /root/reference at build time) or no mpirun")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = str(EXE.parent / "mpilib") + ":" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([str(MPIRUN), "-np", str(np_), str(EXE)] + args, capture_output=True, text=True, env=env, cwd="/tmp",
                       timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "ERROR" not in r.stdout, r.stdout[-2000:]
    return r.stdout


@pytest.mark.parametrize("np_,prec,size,opts,tol", [
    (1, "double", (64, 64, 64), ["-slabs", "-p2p_pl"], 1e-11),
    (2, "double", (96, 48, 80), ["-slabs", "-p2p_pl"], 1e-11),       # radix-3 / radix-5 axes, two ranks
    (4, "double", (128, 64, 32), ["-pencils", "-a2a"], 1e-11),         # pencil reshapes: column-type executors too
    (2, "float", (64, 64, 64), ["-slabs", "-p2p_pl"], 5e-4),
    (2, "double", (64, 64, 64), ["-slabs", "-p2p_pl", "-no-reorder"], 1e-11),  # strided lines (order[1] / order[2] cases)
])
def test_heffte_benchmark_on_the_dfft_backend(gpu, np_, prec, size, opts, tol):
    out = _run(np_, ["stock", prec] + [str(s) for s in size] + opts)
    assert "heFFTe performance test" in out and ("Size:      %dx%dx%d" % size) in out
    err = float(re.search(r"Max error:\s*([0-9.eE+-]+)", out).group(1))
    assert err < tol
    assert float(re.search(r"Time per run:\s*([0-9.eE+-]+)", out).group(1)) > 0
