"""-m gpu: Test_1D / Test_2D, the reference's batched component benchmarks (templateFFT/batchTest/Test_1D.cpp:29-198,
Test_2D.cpp) rebuilt on the C-ABI: same CLI, same printed lines, same CSV schema, one published row of every radix family."""
import math
import os
import re
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "distributedfft_amd" / "lib"
HDR = "X,Y,Z,Buffer,hip_time,GFlops,num_iter,bandwidth,max error"


def _run(exe, args, tmp_path):
    csv = tmp_path / "rows.csv"
    env = dict(os.environ, DFFT_BATCH_CSV=str(csv))
    r = subprocess.run([str(LIB / exe)] + [str(a) for a in args], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    row = csv.read_text().strip().splitlines()[-1].split(",")
    assert len(row) == len(HDR.split(","))
    return r.stdout, row


# input re = i + 1 up to 2^26: the round trip's absolute error scales with that magnitude (the reference's own table shows
# 2.8e-13 ... 1.2e-10 for its radix-5 rows); 1e-6 absolute = 1.5e-14 relative to the largest input
@pytest.mark.parametrize("x", [512, 243, 625, 2187, 3125, 343, 1000, 8192, 15625, 131072])
def test_test_1d_rows_of_every_radix_family(gpu, tmp_path, x):
    out, row = _run("Test_1D", [x, 1, 1, 20, 0], tmp_path)
    assert "1 - FFT + iFFT C2C 1D in double precision LUT" in out and re.search(r"FFT: %dx\d+x1 Buffer: " % x, out)
    y = (1 << 26) // x
    assert int(row[0]) == x and int(row[1]) == y and int(row[2]) == 1 and int(row[6]) == 20
    assert abs(float(row[3]) - x * y * 16 / 2 ** 20) < 1e-2
    gflops = y * 5.0 * x * math.log2(x) / (1e6 * float(row[4]))
    assert abs(float(row[5]) / gflops - 1) < 1e-3 and float(row[5]) > 100
    assert float(row[8]) < 1e-6 and float(re.search(r"Max error: ([0-9.eE+-]+)", out).group(1)) == pytest.approx(float(row[8]), rel=1e-3)


@pytest.mark.parametrize("x,y", [(512, 512), (243, 243), (729, 243), (360, 360), (2048, 128)])
def test_test_2d_rows(gpu, tmp_path, x, y):
    out, row = _run("Test_2D", [x, y, 1, 10, 0], tmp_path)
    assert "1 - FFT + iFFT C2C 2D in double precision LUT" in out
    z = (1 << 26) // (x * y)
    assert [int(row[0]), int(row[1]), int(row[2])] == [x, y, z]
    assert float(row[5]) > 100 and float(row[8]) < 1e-6


def test_unsupported_length_is_refused(gpu, tmp_path):
    r = subprocess.run([str(LIB / "Test_1D"), "8191", "1", "1", "5", "0"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "unsupported length" in r.stderr
