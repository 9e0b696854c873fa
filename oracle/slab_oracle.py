"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's slab 3D C2C FFT and loader for the C restatement.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The product path
(distributedfft_amd -> libdfft_mi355x.so) never does.

Two independent statements of the same mathematics are kept so that they can check each other:
  * `stockham_fft` / `slab_pipeline`: follow the reference stage by stage (files/lines cited per function);
  * `fftn_reference`: numpy.fft.fftn (pocketfft) of the whole array, re-laid-out to the reference's output layout.
Pinning status: pinned against heFFTe's golden vectors bundled with the reference (tests/test_oracle_golden.py) and, on
the GPU box, against the reference's own GPU implementation built from /root/reference (oracle/_ref, see oracle/Makefile).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
from pathlib import Path
from typing import List, Sequence

import numpy as np

HERE = Path(__file__).resolve().parent
C_LIB = HERE / "libslab_oracle.so"


# ---- decomposition (fft_mpi_3d_api.cpp:84-133, 232-316) ------------------------------------------------------------------
def slab_size(n: int, P: int, g: int) -> int:
    blk = -(-n // P)
    return blk if g < P - 1 else n - (P - 1) * blk


def slab_start(n: int, P: int, g: int) -> int:
    return g * (-(-n // P))


def proper_device_num(N: Sequence[int], ini: int, mpi_size: int, mpi_rank: int):
    """getProperDeviceNum, fft_mpi_3d_api.cpp:232-272 (without the clamp to the visible device count)."""
    total, in_rank = ini * mpi_size, ini
    if N[0] % total != 0:
        per = N[0] // total + 1
        total = N[0] // per + (1 if N[0] % per else 0)
        in_rank = total // mpi_size
        if total % mpi_size and mpi_rank < total % mpi_size:
            in_rank += 1
    return total, in_rank


def max_data_count(n0, n1, n2, P, is_last) -> int:
    """getMaxDataCount, fft_mpi_3d_api.cpp:289-316."""
    g = P - 1 if is_last else 0
    return max(slab_size(n0, P, g) * n1 * n2, n0 * slab_size(n1, P, g) * n2)


def exchange_layout(n0, n1, n2, P, g, direction):
    """tInfo counts/offsets in elements, fft_mpi_3d_api.cpp:84-133 and receive offsets :618-625."""
    xl, yl = -(-n0 // P), -(-n1 // P)
    sc, so, rc, ro = [], [], [], []
    for q in range(P):
        if direction > 0:
            sc.append(slab_size(n0, P, g) * slab_size(n1, P, q) * n2)
            so.append(q * slab_size(n0, P, g) * yl * n2)
            rc.append(slab_size(n0, P, q) * slab_size(n1, P, g) * n2)
            ro.append(q * xl * slab_size(n1, P, g) * n2)
        else:
            sc.append(slab_size(n0, P, q) * slab_size(n1, P, g) * n2)
            so.append(q * xl * slab_size(n1, P, g) * n2)
            rc.append(slab_size(n0, P, g) * slab_size(n1, P, q) * n2)
            ro.append(q * slab_size(n0, P, g) * yl * n2)
    return sc, so, rc, ro


# ---- inputs -----------------------------------------------------------------------------------------------
def driver_input(N, P: int, g: int, dtype=np.complex128) -> np.ndarray:
    """I-ref: re = im = global linear index (fftSpeed3d_c2c.cpp:56-63); returns device g's [xs][N1][N2] slab."""
    n0, n1, n2 = N
    first = slab_start(n0, P, g) * n1 * n2
    cnt = slab_size(n0, P, g) * n1 * n2
    v = np.arange(first, first + cnt, dtype=np.float64)
    return (v + 1j * v).astype(dtype).reshape(slab_size(n0, P, g), n1, n2)


def minstd_uniform(count: int, seed: int = 4242) -> np.ndarray:
    """I-rand: std::minstd_rand(seed) + uniform_real_distribution<double>(0,1) draw order = linear index
    (heFFTe test_fft3d.h:20-28).  libstdc++'s generate_canonical<double,53> takes 2 draws per double for minstd
    (range 2^31-2): value = (d0 + d1 * R) / R^2 with R = 2147483646, d = x - 1."""
    a, m = 48271, 2147483647
    x = seed % m
    R = float(m - 1)
    out = np.empty(count, dtype=np.float64)
    for i in range(count):
        x = (x * a) % m
        d0 = x - 1
        x = (x * a) % m
        d1 = x - 1
        v = (d0 + d1 * R) / (R * R)
        out[i] = v if v < 1.0 else math.nextafter(1.0, 0.0)
    return out


def random_input(N, seed: int = 1234, dtype=np.complex128) -> np.ndarray:
    """Full [N0][N1][N2] array of uniform [-1,1) re/im from numpy's PCG64 (fast path for large parity cases)."""
    rng = np.random.default_rng(seed)
    n = int(np.prod(N))
    x = rng.uniform(-1.0, 1.0, size=2 * n).view(np.complex128)
    return x.astype(dtype).reshape(N)


# ---- 1D transform: Stockham autosort, as templateFFT emits it --------------------------------------------------------------
def radix_plan(n: int) -> List[int]:
    """fold 2s into 8, 4, 2, then 3, 5, 7; largest first (templateFFT.cpp:4540-4588)."""
    plan, m = [], n
    for r in (8, 4, 2, 3, 5, 7) + tuple(range(11, 62, 2)):  # > 7: oracle-only by-definition stages
        while m % r == 0:
            plan.append(r)
            m //= r
    if m != 1:
        raise ValueError(f"length {n} has a prime factor > 61")
    return plan


def stockham_fft(x: np.ndarray, direction: int = 1, plan: Sequence[int] | None = None) -> np.ndarray:
    """Length-n C2C FFT of the last axis by radix stages (templateFFT.cpp:1871-2045 appendRadixStage/appendRadixShuffle):
    stage with radix r and stride S = product of earlier radices: butterfly j < n/r reads x[j + k n/r], multiplies by
    e^{-+2 pi i k (j mod S)/(S r)} (:337-341), does a length-r DFT and scatters to (j - j%S) r + j%S + q S."""
    x = np.asarray(x)
    n = x.shape[-1]
    plan = radix_plan(n) if plan is None else list(plan)
    assert int(np.prod(plan)) == n if plan else n == 1
    sign = -1.0 if direction > 0 else 1.0
    cur = x.astype(np.complex128, copy=True)
    S = 1
    for r in plan:
        nb = n // r
        j = np.arange(nb)
        jm = j % S
        k = np.arange(r)
        tw = np.exp(sign * 2j * np.pi * np.outer(k, jm) / (S * r))                # [r][nb]
        u = cur.reshape(cur.shape[:-1] + (r, nb)) * tw                             # x[j + k*nb]
        dft = np.exp(sign * 2j * np.pi * np.outer(k, k) / r)                       # [q][k]
        v = np.einsum("qk,...kj->...qj", dft, u)
        nxt = np.empty_like(cur)
        base = (j - jm) * r + jm
        for q in range(r):
            nxt[..., base + q * S] = v[..., q, :]
        cur = nxt
        S *= r
    return cur


# ---- the pipeline, stage by stage ------------------------------------------------------------------------------------------
def t0_fft_yz(slab: np.ndarray, direction: int) -> np.ndarray:
    """fftZY, fft_mpi_3d_api.cpp:466-522: per X-plane, Z pass (contiguous) then Y pass (stride N2)."""
    z = stockham_fft(slab, direction)
    return np.swapaxes(stockham_fft(np.swapaxes(z, 1, 2), direction), 1, 2)


def t1_pack(slab: np.ndarray, P: int) -> np.ndarray:
    """slab_local_transpose_z_to_x_uneven_forward_optimized, kernel_func.cpp:73-86 -> flat packed buffer."""
    xs, n1, n2 = slab.shape
    yl = -(-n1 // P)
    out = np.zeros(max(xs * n1 * n2, P * xs * yl * n2), dtype=slab.dtype)
    for d in range(P):
        y0, yw = slab_start(n1, P, d), slab_size(n1, P, d)
        blk = slab[:, y0:y0 + yw, :].reshape(-1)
        off = d * xs * yl * n2
        out[off:off + blk.size] = blk
    return out


def t1_unpack(packed: np.ndarray, xs: int, n1: int, n2: int, P: int) -> np.ndarray:
    """..._backward_optimized, kernel_func.cpp:88-100."""
    yl = -(-n1 // P)
    slab = np.empty((xs, n1, n2), dtype=packed.dtype)
    for d in range(P):
        y0, yw = slab_start(n1, P, d), slab_size(n1, P, d)
        off = d * xs * yl * n2
        slab[:, y0:y0 + yw, :] = packed[off:off + xs * yw * n2].reshape(xs, yw, n2)
    return slab


def t2_exchange(send: List[np.ndarray], N, P: int, direction: int) -> List[np.ndarray]:
    """slabAlltoall, fft_mpi_3d_api.cpp:610-672: chunk(src -> dst) lands at dst's receive offset for src."""
    n0, n1, n2 = N
    recv = [np.zeros(max_data_count(n0, n1, n2, P, g == P - 1), dtype=send[0].dtype) for g in range(P)]
    lay = [exchange_layout(n0, n1, n2, P, g, direction) for g in range(P)]
    for g in range(P):
        sc, so, _, _ = lay[g]
        for d in range(P):
            _, _, rc, ro = lay[d]
            assert sc[d] == rc[g]
            recv[d][ro[g]:ro[g] + rc[g]] = send[g][so[d]:so[d] + sc[d]]
    return recv


def t3_fft_x(recv: np.ndarray, n0: int, ys: int, n2: int, direction: int) -> np.ndarray:
    """fftX forward, fft_mpi_3d_api.cpp:524-552: transpose [N0][ys*N2] -> [ys*N2][N0] (kernels_201.cpp:56-57), FFT of N0."""
    a = recv[:n0 * ys * n2].reshape(n0, ys * n2).T
    return stockham_fft(np.ascontiguousarray(a), direction).reshape(ys, n2, n0)


def slab_pipeline_forward(x: np.ndarray, P: int) -> List[np.ndarray]:
    """Forward transform of the full [N0][N1][N2] array over P emulated devices -> per-device [ys][N2][N0]."""
    n0, n1, n2 = x.shape
    send = []
    for g in range(P):
        slab = x[slab_start(n0, P, g):slab_start(n0, P, g) + slab_size(n0, P, g)]
        send.append(t1_pack(t0_fft_yz(slab, +1), P))
    recv = t2_exchange(send, (n0, n1, n2), P, +1)
    return [t3_fft_x(recv[d], n0, slab_size(n1, P, d), n2, +1) for d in range(P)]


def slab_pipeline_backward(y: List[np.ndarray], N, P: int) -> List[np.ndarray]:
    """Backward (unnormalised) of per-device [ys][N2][N0] -> per-device [xs][N1][N2] (fft_mpi_3d_api.cpp:203-212)."""
    n0, n1, n2 = N
    send = []
    for d in range(P):
        ys = slab_size(n1, P, d)
        a = stockham_fft(y[d].reshape(ys * n2, n0), -1)              # inverse X FFT in place
        buf = np.zeros(max_data_count(n0, n1, n2, P, d == P - 1), dtype=a.dtype)
        buf[:n0 * ys * n2] = np.ascontiguousarray(a.T).reshape(-1)  # transpose 120 -> [x][ys][N2]
        send.append(buf)
    recv = t2_exchange(send, N, P, -1)
    out = []
    for g in range(P):
        slab = t1_unpack(recv[g], slab_size(n0, P, g), n1, n2, P)
        yz = np.swapaxes(stockham_fft(np.swapaxes(slab, 1, 2), -1), 1, 2)
        out.append(stockham_fft(yz, -1))
    return out


# ---- independent statement ------------------------------------------------------------------------------------------------
def fftn_reference(x: np.ndarray, P: int, direction: int = 1) -> List[np.ndarray]:
    """numpy.fft.fftn of the whole array, cut to what device d holds after the forward transform:
    out_d[yy][z][kx] = F[kx][d*yl+yy][z]  (SURVEY Appendix B)."""
    n0, n1, n2 = x.shape
    F = np.fft.fftn(x.astype(np.complex128)) if direction > 0 else np.fft.ifftn(x.astype(np.complex128)) * x.size
    return [np.ascontiguousarray(F[:, slab_start(n1, P, d):slab_start(n1, P, d) + slab_size(n1, P, d), :].transpose(1, 2, 0))
            for d in range(P)]


def driver_error(inp: np.ndarray, roundtrip: np.ndarray, N) -> float:
    """The driver's printed metric, fftSpeed3d_c2c.cpp:84-91: max |x - ifft(fft(x))/N| / 1e7."""
    n = float(N[0]) * N[1] * N[2]
    d = inp - roundtrip / n
    return float(np.max(np.abs(d))) / 1e7


# ---- C restatement (oracle/slab_oracle.c) ------------------------------------------------------------------------------------
def build_c(force: bool = False) -> Path:
    src = HERE / "slab_oracle.c"
    if force or not C_LIB.exists() or C_LIB.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["gcc", "-O2", "-std=c99", "-fPIC", "-shared", "-D_POSIX_C_SOURCE=199309L", "-o", str(C_LIB), str(src),
                        "-lm"], check=True)
    return C_LIB


_clib = None


def c_lib() -> C.CDLL:
    global _clib
    if _clib is None:
        lib = C.CDLL(str(build_c()))
        dp = C.POINTER(C.c_double)
        lib.oracle_fft1d.restype = C.c_int
        lib.oracle_fft1d.argtypes = [dp, C.c_int, C.c_long, C.c_int]
        lib.oracle_driver_input.restype = None
        lib.oracle_driver_input.argtypes = [dp, C.c_long, C.c_long, C.c_long, C.c_int, C.c_int]
        lib.oracle_slab_fft3d.restype = C.c_int
        lib.oracle_slab_fft3d.argtypes = [dp, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp]
        _clib = lib
    return _clib


def c_fft1d(x: np.ndarray, direction: int = 1) -> np.ndarray:
    a = np.ascontiguousarray(x, dtype=np.complex128).copy()
    n = a.shape[-1]
    rc = c_lib().oracle_fft1d(a.view(np.float64).ctypes.data_as(C.POINTER(C.c_double)), n, a.size // n, direction)
    if rc:
        raise ValueError(f"oracle_fft1d: unsupported length {n}")
    return a


def c_slab_fft3d(x: np.ndarray, N, P: int, direction: int = 1):
    """Returns (flat output, stage seconds[4]).  Forward: x = [N0][N1][N2], output = concat_d [ys_d][N2][N0]."""
    n0, n1, n2 = N
    a = np.ascontiguousarray(x, dtype=np.complex128).reshape(-1)
    out = np.empty_like(a)
    st = (C.c_double * 4)()
    dp = C.POINTER(C.c_double)
    rc = c_lib().oracle_slab_fft3d(a.view(np.float64).ctypes.data_as(dp), out.view(np.float64).ctypes.data_as(dp), n0, n1, n2, P,
                                   direction, st)
    if rc:
        raise ValueError(f"oracle_slab_fft3d failed ({rc})")
    return out, list(st)


def split_forward_output(flat: np.ndarray, N, P: int) -> List[np.ndarray]:
    n0, n1, n2 = N
    res, off = [], 0
    for d in range(P):
        ys = slab_size(n1, P, d)
        res.append(flat[off:off + ys * n2 * n0].reshape(ys, n2, n0))
        off += ys * n2 * n0
    return res
