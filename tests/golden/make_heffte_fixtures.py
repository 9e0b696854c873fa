"""Regenerate the heFFTe golden fixtures (run in the build container, where /root/reference exists).

Builds oracle/_ref/heffte_dump (heFFTe 2.1.0 stock CPU backend compiled from the reference's bundled sources, see
oracle/Makefile and oracle/ref/heffte_dump.cpp) and stores the forward outputs for seed-4242 worlds
(test_fft3d.h:20-28 input recipe).  Inputs are not stored: oracle.slab_oracle.minstd_uniform reproduces them bit for bit
(checked here).  Shapes are given in OUR order (N0 slowest .. N2 fastest); heFFTe's index 0 is the fastest dimension.
"""
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import slab_oracle as so  # noqa: E402

SHAPES = [(16, 12, 8), (6, 9, 10), (32, 32, 32)]


def main():
    subprocess.run(["make", "-C", str(ROOT / "oracle"), "_ref/heffte_dump"], check=True, capture_output=True)
    exe = ROOT / "oracle" / "_ref" / "heffte_dump"
    for N in SHAPES:
        with tempfile.TemporaryDirectory() as td:
            prefix = str(Path(td) / "hf")
            subprocess.run([str(exe), str(N[2]), str(N[1]), str(N[0]), prefix], check=True, cwd=td)
            x = np.fromfile(prefix + ".in", dtype=np.complex128)
            y = np.fromfile(prefix + ".out", dtype=np.complex128).reshape(N)
        u = so.minstd_uniform(x.size)
        assert np.array_equal(x.real, u) and not x.imag.any(), "minstd port no longer matches libstdc++"
        out = Path(__file__).parent / f"heffte_stock_fwd_{N[0]}x{N[1]}x{N[2]}.npy"
        np.save(out, y)
        print("wrote", out, y.shape)


if __name__ == "__main__":
    main()
