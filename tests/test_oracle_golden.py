"""Pin the oracle (oracle/slab_oracle.{c,py}) before trusting it: every golden vector / known-answer test the reference
holds for this path plus fixtures produced by the reference's bundled heFFTe stock backend.
CPU only."""
from pathlib import Path

import numpy as np
import pytest

from oracle import slab_oracle as so

GOLDEN = Path(__file__).parent / "golden"


@pytest.mark.parametrize("n", [2, 3, 4, 5, 8, 9, 11, 16, 25, 27, 64, 81, 128, 243, 512, 625, 768, 1024])
def test_impulse_gives_all_ones(n):
    """heFFTe test_units_stock.cpp:188-204: FFT of a unit impulse is exactly all ones."""
    x = np.zeros((1, n), dtype=np.complex128)
    x[0, 0] = 1
    for got in (so.c_fft1d(x), so.stockham_fft(x)):
        assert np.abs(got - 1.0).max() < 1e-14


def test_dft_of_1_to_11_matches_heffte_constants():
    """test_units_stock.cpp:230-253: DFT of [1..11] = 66, then -5.5 +- i*{18.73.., 8.558.., 4.765.., 2.511.., 0.790..}."""
    imag = [18.731279813890875, 8.55816705136493, 4.765777128986846, 2.5117658384695547, 0.790780616972353]
    ref = np.empty(11, dtype=np.complex128)
    ref[0] = 66
    for i in range(1, 6):
        ref[i] = complex(-5.5, imag[i - 1])
        ref[11 - i] = complex(-5.5, -imag[i - 1])
    x = np.arange(1, 12, dtype=np.float64)[None, :] + 0j
    for got in (so.c_fft1d(x)[0], so.stockham_fft(x)[0], np.fft.fft(x)[0]):
        assert np.abs(got - ref).max() < 1e-13


@pytest.mark.parametrize("n", [16, 9, 64, 20, 48, 125])
def test_1_to_n_against_naive_dft_and_roundtrip(n):
    """test_units_stock.cpp:206-225: input [1..N] against the DFT by definition; backward/N returns the input."""
    x = np.arange(1, n + 1, dtype=np.float64) + 0j
    k = np.arange(n)
    naive = (x[None, :] * np.exp(-2j * np.pi * np.outer(k, k) / n)).sum(axis=1)
    fwd = so.c_fft1d(x[None, :])[0]
    assert np.abs(fwd - naive).max() / np.abs(naive).max() < 1e-14
    back = so.c_fft1d(fwd[None, :], -1)[0] / n
    assert np.abs(back - x).max() < 1e-12
    assert np.abs(so.stockham_fft(x[None, :])[0] - naive).max() / np.abs(naive).max() < 1e-14


def test_heffte_2x3x4_box_pen_and_paper_vectors():
    """test_units_nompi.cpp:93-206: box 2x3x4 (index 0 fastest) holding 1..24; 1D transforms along each dimension."""
    data = np.arange(1, 25, dtype=np.float64).reshape(4, 3, 2) + 0j  # [k][j][i], i fastest
    # dimension 0 (size 2): (3 + 2*idx, -1)
    f0 = so.c_fft1d(data.reshape(12, 2)).reshape(-1)
    exp0 = np.empty(24, dtype=np.complex128)
    exp0[0::2] = 3 + 2 * np.arange(0, 24, 2)
    exp0[1::2] = -1
    assert np.abs(f0 - exp0).max() < 1e-13
    # dimension 1 (size 3): ((2j+i+1)*9 - 6i, -3 +- 1.73205080756888 i)
    f1 = np.swapaxes(so.c_fft1d(np.ascontiguousarray(np.swapaxes(data, 1, 2))), 1, 2).reshape(-1)
    exp1 = np.empty(24, dtype=np.complex128)
    for j in range(4):
        for i in range(2):
            exp1[6 * j + i] = (2 * j + i + 1) * 9.0 - i * 6.0
            exp1[6 * j + i + 2] = complex(-3.0, 1.73205080756888)
            exp1[6 * j + i + 4] = complex(-3.0, -1.73205080756888)
    assert np.abs(f1 - exp1).max() < 1e-13
    # dimension 2 (size 4): (40 + 4 idx, -12+12i, -12, -12-12i)
    f2 = np.moveaxis(so.c_fft1d(np.ascontiguousarray(np.moveaxis(data, 0, 2))), 2, 0).reshape(-1)
    exp2 = np.empty(24, dtype=np.complex128)
    for i in range(6):
        exp2[i] = 40.0 + 4 * i
        exp2[i + 6] = complex(-12.0, 12.0)
        exp2[i + 12] = -12.0
        exp2[i + 18] = complex(-12.0, -12.0)
    assert np.abs(f2 - exp2).max() < 1e-13
    # the same box through the whole slab pipeline, every P the decomposition allows, vs fftn
    x = np.ascontiguousarray(data.transpose(0, 1, 2))  # treat as [N0=4][N1=3][N2=2]
    for P in (1, 2):
        flat, _ = so.c_slab_fft3d(x, (4, 3, 2), P, +1)
        ref = np.concatenate([r.reshape(-1) for r in so.fftn_reference(x, P)])
        assert np.abs(flat - ref).max() < 1e-12


@pytest.mark.parametrize("N", [(16, 12, 8), (6, 9, 10), (32, 32, 32)])
@pytest.mark.parametrize("P", [1, 2, 4])
def test_heffte_stock_fixture_seed_4242(N, P):
    """Fixture from the reference's bundled heFFTe stock backend (tests/golden/make_heffte_fixtures.py), input recipe and
    tolerance of heFFTe's own fft3d test: minstd_rand(4242) world, 1e-11 (test_fft3d.h:20-28, test_common.h:136-140)."""
    n0, n1, n2 = N
    if so.slab_size(n0, P, P - 1) < 1 or so.slab_size(n1, P, P - 1) < 1:
        pytest.skip("empty last slab")
    golden = np.load(GOLDEN / f"heffte_stock_fwd_{n0}x{n1}x{n2}.npy")        # [kx][ky][kz]
    x = so.minstd_uniform(n0 * n1 * n2).reshape(N) + 0j
    want = [np.ascontiguousarray(golden[:, so.slab_start(n1, P, d):so.slab_start(n1, P, d) + so.slab_size(n1, P, d), :]
                                 .transpose(1, 2, 0)) for d in range(P)]
    flat, _ = so.c_slab_fft3d(x, N, P, +1)
    c_out = so.split_forward_output(flat, N, P)
    py_out = so.slab_pipeline_forward(x, P)
    np_out = so.fftn_reference(x, P)
    for d in range(P):
        assert np.abs(c_out[d] - want[d]).max() < 1e-11
        assert np.abs(py_out[d] - want[d]).max() < 1e-11
        assert np.abs(np_out[d] - want[d]).max() < 1e-11
    # backward of the golden output returns N * input (heFFTe's backward is unnormalised as well)
    back, _ = so.c_slab_fft3d(np.concatenate([w.reshape(-1) for w in want]), N, P, -1)
    assert np.abs(back.reshape(N) / (n0 * n1 * n2) - x).max() < 1e-12


@pytest.mark.parametrize("N,P", [((8, 6, 4), 1), ((8, 6, 4), 2), ((10, 10, 4), 4), ((16, 16, 16), 4), ((7, 5, 3), 3),
                                 ((24, 10, 12), 4), ((25, 10, 16), 4)])
def test_three_statements_agree(N, P):
    """C restatement == numpy stage-by-stage restatement == numpy.fftn, forward and backward, even and uneven slabs."""
    x = so.random_input(N, seed=sum(N) + P)
    ref = so.fftn_reference(x, P)
    py = so.slab_pipeline_forward(x, P)
    flat, st = so.c_slab_fft3d(x, N, P, +1)
    cc = so.split_forward_output(flat, N, P)
    scale = max(np.abs(r).max() for r in ref)
    for d in range(P):
        assert np.abs(py[d] - ref[d]).max() / scale < 1e-13
        assert np.abs(cc[d] - ref[d]).max() / scale < 1e-13
    back = np.concatenate(so.slab_pipeline_backward(py, N, P), axis=0)
    assert np.abs(back / np.prod(N) - x).max() < 1e-12
    bflat, _ = so.c_slab_fft3d(flat, N, P, -1)
    assert np.abs(bflat.reshape(N) / np.prod(N) - x).max() < 1e-12
    assert len(st) == 4 and all(t >= 0 for t in st)


def test_driver_input_and_error_metric():
    """fftSpeed3d_c2c.cpp:56-63 input and :84-91 error formula on a perfect round trip."""
    N, P = (8, 4, 4), 2
    slabs = [so.driver_input(N, P, g) for g in range(P)]
    full = np.concatenate(slabs, axis=0)
    assert full.real[1, 2, 3] == (1 * 4 + 2) * 4 + 3 and np.array_equal(full.real, full.imag)
    import ctypes as C
    buf = np.empty(slabs[1].size, dtype=np.complex128)
    so.c_lib().oracle_driver_input(buf.view(np.float64).ctypes.data_as(C.POINTER(C.c_double)), *N, P, 1)
    assert np.array_equal(buf.reshape(slabs[1].shape), slabs[1])
    fwd = so.slab_pipeline_forward(full, P)
    back = so.slab_pipeline_backward(fwd, N, P)
    err = max(so.driver_error(slabs[g].reshape(-1), back[g].reshape(-1), N) for g in range(P))
    assert err < 1e-15
