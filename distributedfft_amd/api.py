"""Host-side mirror of the reference's distributed-FFT API over the C-ABI (include/dfft.h).

Function names and argument meaning follow /root/reference/3dmpifft_opt/include/fft_mpi_3d_api.h:68-86 so that tests read
like the reference's driver (fftSpeed3d_c2c.cpp): fft_mpi_init -> fft_mpi_plan_dft_c2c_3d -> fft_mpi_execute_dft_3d_c2c ->
fft_mpi_destroy_plan.  torch is used only to own device memory and to select the device; all arithmetic happens in the
HIP kernels of libdfft_mi355x.so.  There is no CPU path here.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

from . import _lib as L
from ._lib import (BACKWARD, EXEC_ASYNC, EXEC_NO_TIMING, EXEC_PRINT, EXEC_SYNC_STAGES, F32, F64, FORWARD, PLAN_DEFAULT,  # noqa: F401
                   PLAN_INPUT_FROM_IN, PLAN_NATURAL, PLAN_OVERLAP, PLAN_UNFUSED, DfftError)


def _ll3(N: Sequence[int]):
    return (C.c_longlong * 3)(int(N[0]), int(N[1]), int(N[2]))


# ---- host-only slab bookkeeping (usable without a GPU) ------------------------------------------------------------------
def get_proper_device_num(N, ini_devices_in_rank: int, mpi_size: int, mpi_rank: int, real_devices: int = -1) -> Tuple[int, int]:
    """getProperDeviceNum, fft_mpi_3d_api.cpp:232-272 -> (newDeviceCount, newDeviceCountInNode)."""
    lib = L.load()
    tot, inr = C.c_int(), C.c_int()
    L.check(lib.dfft_proper_device_count(_ll3(N), ini_devices_in_rank, mpi_size, mpi_rank, real_devices, C.byref(tot),
                                         C.byref(inr)), "dfft_proper_device_count")
    return tot.value, inr.value


def get_data_count(N, total_devices: int, global_idx: int) -> int:
    """getDataCountForNode, fft_mpi_3d_api.cpp:274-287 (per device)."""
    return int(L.load().dfft_local_count(_ll3(N), total_devices, global_idx))


def get_max_data_count(n0: int, n1: int, n2: int, total_devices: int, is_last: bool) -> int:
    """getMaxDataCount, fft_mpi_3d_api.cpp:289-316."""
    return int(L.load().dfft_max_count(n0, n1, n2, total_devices, 1 if is_last else 0))


@dataclass
class ExchangeLayout:
    scount: List[int]
    soffset: List[int]
    rcount: List[int]
    roffset: List[int]


def exchange_layout(n0, n1, n2, total_devices: int, global_idx: int, direction: int) -> ExchangeLayout:
    """tInfo of fft_mpi_plan_dft_c2c_3d, fft_mpi_3d_api.cpp:84-133 (elements)."""
    lib = L.load()
    arrs = [(C.c_longlong * total_devices)() for _ in range(4)]
    L.check(lib.dfft_exchange_layout(n0, n1, n2, total_devices, global_idx, direction, *arrs), "dfft_exchange_layout")
    return ExchangeLayout(*[list(a) for a in arrs])


def exchange_part_layout(n0, n1, n2, total_devices: int, global_idx: int, part_planes: int, part: int, ycuts: int = 1,
                         ycut: int = -1, direction: int = FORWARD):
    """Messages of one piece of the overlapped exchange: list of (peer, soffset, scount, roffset, rcount)."""
    lib = L.load()
    cap = total_devices * max(1, ycuts)
    peer = (C.c_int * cap)()
    arrs = [(C.c_longlong * cap)() for _ in range(4)]
    n = lib.dfft_exchange_part_layout(n0, n1, n2, total_devices, global_idx, direction, part_planes, part, ycuts, ycut, cap,
                                      peer, *arrs)
    if n < 0:
        L.check(n, "dfft_exchange_part_layout")
    return [(peer[i], arrs[0][i], arrs[1][i], arrs[2][i], arrs[3][i]) for i in range(n)]


def local_size(n0, n1, n2, total_devices: int, global_idx: int) -> Tuple[int, int, int, int]:
    """(local_n0, local_0_start, local_n1, local_1_start) -- fft_mpi_local_size_3d (declared, fft_mpi_3d_api.h:73)."""
    lib = L.load()
    v = [C.c_longlong() for _ in range(4)]
    L.check(lib.dfft_local_size(n0, n1, n2, total_devices, global_idx, *[C.byref(x) for x in v]), "dfft_local_size")
    return tuple(int(x.value) for x in v)


def fft_mpi_init(N, ini_devices_in_rank: int, mpi_size: int = 1, mpi_rank: int = 0, real_devices: int = -1):
    """fft_mpi_init, fft_mpi_3d_api.cpp:3-39 -> (newDeviceCount, newDeviceCountInNode, dataCountInNode[])."""
    tot, inr = get_proper_device_num(N, ini_devices_in_rank, mpi_size, mpi_rank, real_devices)
    import math
    first = mpi_rank * math.ceil(tot / mpi_size)
    counts = [get_data_count(N, tot, first + i) for i in range(inr)]
    return tot, inr, counts


# ---- communicators -----------------------------------------------------------------------------------------------------
def device_pci_bus_id(device: int = -1) -> str:
    """PCI address of a HIP device (-1: the current one) -- dfft_device_pci_bus_id."""
    buf = C.create_string_buffer(64)
    L.check(L.load().dfft_device_pci_bus_id(device, buf, 64), "dfft_device_pci_bus_id")
    return buf.value.decode()


class Comm:
    def __init__(self, handle, kind: str, size: int):
        self.handle, self.kind, self.size = handle, kind, size

    @staticmethod
    def local(total_devices: int) -> "Comm":
        h = C.c_void_p()
        L.check(L.load().dfft_comm_create_local(total_devices, C.byref(h)), "dfft_comm_create_local")
        return Comm(h, "local", total_devices)

    @staticmethod
    def rccl_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        L.check(L.load().dfft_rccl_unique_id(buf), "dfft_rccl_unique_id")
        return buf.raw

    @staticmethod
    def rccl(unique_id: bytes, total_devices: int, global_idx: int) -> "Comm":
        assert len(unique_id) == 128
        h = C.c_void_p()
        L.check(L.load().dfft_comm_create_rccl(unique_id, total_devices, global_idx, C.byref(h)), "dfft_comm_create_rccl")
        return Comm(h, "rccl", total_devices)

    @staticmethod
    def ipc(total_devices: int, global_idx: int, async_exchange: bool = False) -> "Comm":
        """One process per device without RCCL: hipIpc-shared receive buffers, device-to-device pushes, barriers over the
        dfft_boot_* rendezvous (DFFT_RANK/... or torchrun's RANK/WORLD_SIZE/MASTER_ADDR/MASTER_PORT)."""
        h = C.c_void_p()
        L.check(L.load().dfft_comm_create_ipc(total_devices, global_idx, 1 if async_exchange else 0, C.byref(h)),
                "dfft_comm_create_ipc")
        return Comm(h, "ipc-async" if async_exchange else "ipc", total_devices)

    def info(self) -> dict:
        """{'kind', 'size', 'rank', 'device'} as the transport reports them (dfft_comm_info)."""
        k, s, r, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        L.check(L.load().dfft_comm_info(self.handle, C.byref(k), C.byref(s), C.byref(r), C.byref(d)), "dfft_comm_info")
        return {"kind": ("local", "rccl", "ipc", "ipc-async")[k.value], "size": s.value, "rank": r.value, "device": d.value}

    def destroy(self):
        if self.handle:
            L.load().dfft_comm_destroy(self.handle)
            self.handle = None


# ---- plans ----------------------------------------------------------------------------------------------------------------
def _dtype_code(t) -> int:
    import torch
    if t.dtype == torch.complex128:
        return F64
    if t.dtype == torch.complex64:
        return F32
    raise TypeError("dfft buffers must be torch.complex128 or torch.complex64 device tensors")


class Plan:
    """fft_mpi_3d_plan (fft_mpi_3d_api.h:11-66): owns bufferDev1; `in`/`out` tensors stay owned by the caller."""

    def __init__(self, n0, n1, n2, inp, out, comm: Optional[Comm], global_idx: int, total_devices: int, direction: int,
                 flags: int = PLAN_DEFAULT):
        import torch
        lib = L.load()
        if not inp.is_cuda:
            raise DfftError(L.ENOGPU, "Plan", "buffers must live on a HIP device (no CPU fallback)")
        self.N = (int(n0), int(n1), int(n2))
        self.dtype = _dtype_code(inp)
        self.direction = direction
        self.total_devices, self.global_idx = total_devices, global_idx
        self.max_count = get_max_data_count(n0, n1, n2, total_devices, global_idx == total_devices - 1)
        if inp.numel() < self.max_count or (out is not None and out.numel() < self.max_count):
            raise ValueError(f"in/out must hold getMaxDataCount = {self.max_count} elements")
        self._in, self._out, self._comm = inp, out, comm  # keep alive
        self.handle = C.c_void_p()
        torch.cuda.synchronize(inp.device)  # plan creation copies `in` with a blocking hipMemcpy on the null stream
        with torch.cuda.device(inp.device):
            L.check(lib.dfft_plan_create(C.byref(self.handle), n0, n1, n2, self.dtype, direction, inp.data_ptr(),
                                         out.data_ptr() if out is not None else None,
                                         comm.handle if comm is not None else None, global_idx, total_devices, flags),
                    "dfft_plan_create")
        self.device = inp.device

    @property
    def bufferDev1(self) -> int:
        return int(L.load().dfft_plan_buffer1(self.handle))

    @property
    def stream(self) -> int:
        return int(L.load().dfft_plan_stream(self.handle) or 0)

    def load_input(self, src) -> None:
        """hipMemcpy(plan->bufferDev1, data, ...) as the reference driver does (fftSpeed3d_c2c.cpp:78)."""
        import torch
        n = src.numel()
        assert n <= self.max_count and _dtype_code(src) == self.dtype
        view = self.buffer1_tensor(n)
        view.copy_(src.reshape(-1))
        torch.cuda.synchronize(self.device)

    def buffer1_tensor(self, count: Optional[int] = None):
        """A torch view of bufferDev1 (no copy), via __cuda_array_interface__."""
        import torch
        count = self.max_count if count is None else count
        tdtype = torch.complex128 if self.dtype == F64 else torch.complex64
        comp = "<c16" if self.dtype == F64 else "<c8"

        class _Raw:
            pass

        r = _Raw()
        r.__cuda_array_interface__ = {"shape": (count,), "typestr": comp, "data": (self.bufferDev1, False), "version": 2}
        t = torch.as_tensor(r, device=self.device)
        assert t.dtype == tdtype
        return t

    def execute(self, flags: int = EXEC_ASYNC) -> None:
        """fft_mpi_execute_dft_3d_c2c (fft_mpi_3d_api.cpp:181-214)."""
        import torch
        with torch.cuda.device(self.device):
            L.check(L.load().dfft_execute(self.handle, flags), "dfft_execute")

    def sync(self) -> None:
        L.check(L.load().dfft_plan_sync(self.handle), "dfft_plan_sync")

    def tune(self) -> None:
        """Plan-time measurement (dfft_plan_tune): times the X-pass kernel alone on a few candidate allocations of the plan's
        internal hand-over buffer and keeps the fastest; the probe launches leave garbage in the result buffer."""
        import torch
        with torch.cuda.device(self.device):
            L.check(L.load().dfft_plan_tune(self.handle), "dfft_plan_tune")

    def describe(self) -> str:
        """How this plan executes (dfft_plan_describe): pipeline, YZ stage form, chunk geometry, hand-over buffer, rotation."""
        buf = C.create_string_buffer(512)
        L.check(L.load().dfft_plan_describe(self.handle, buf, 512), "dfft_plan_describe")
        return buf.value.decode()

    def tune_report(self) -> dict:
        """{'candidates_ms': [...], 'kept': i, 'kept_retimed_ms': t} of the last tune() (dfft_plan_tune_report)."""
        ms = (C.c_double * 128)()
        kept, fin = C.c_int(-1), C.c_double(0.0)
        n = L.load().dfft_plan_tune_report(self.handle, 128, ms, C.byref(kept), C.byref(fin))
        if n < 0:
            L.check(n, "dfft_plan_tune_report")
        return {"candidates_ms": [round(ms[i], 4) for i in range(min(n, 128))], "kept": kept.value, "kept_retimed_ms": round(fin.value, 4)}

    def set_scale(self, s: float) -> None:
        """Multiply the result of every later execute by s (folded into the X-pass kernel; 1.0 = the reference's
        un-normalised transform)."""
        L.check(L.load().dfft_plan_set_scale(self.handle, float(s)), "dfft_plan_set_scale")

    def stage_times(self) -> List[float]:
        t = (C.c_double * 4)()
        L.check(L.load().dfft_stage_times(self.handle, t), "dfft_stage_times")
        return list(t)

    def kernel_times(self) -> List[float]:
        """Seconds spent in the Z-row, Y-column and X-column FFT kernels of the last ASYNC execute (HIP events)."""
        t = (C.c_double * 3)()
        L.check(L.load().dfft_kernel_times(self.handle, t), "dfft_kernel_times")
        return list(t)

    def destroy(self) -> None:
        """dfft_plan_destroy.  Its return code is the last place a failure of an execute nobody synchronised THROUGH THE LIBRARY can
        surface (a caller that waits with torch.cuda.synchronize() instead of sync()): raised here, after the handle is gone."""
        if self.handle:
            h, self.handle = self.handle, None
            L.check(L.load().dfft_plan_destroy(h), "dfft_plan_destroy")

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def fft_mpi_plan_dft_c2c_3d(n0, n1, n2, inp, out, comm, global_idx, total_devices, direction, flags=PLAN_DEFAULT) -> Plan:
    return Plan(n0, n1, n2, inp, out, comm, global_idx, total_devices, direction, flags)


def fft_mpi_execute_dft_3d_c2c(plan: Plan, flags: int = EXEC_SYNC_STAGES) -> None:
    plan.execute(flags)
    plan.sync()


def fft_mpi_destroy_plan(plan: Plan) -> None:
    plan.destroy()


# ---- batched 1D building blocks ---------------------------------------------------------------------------------------------
def fft1d_rows(x, direction: int = FORWARD, out=None):
    """Length-n FFT of every contiguous row of a (batch, n) complex device tensor."""
    import torch
    assert x.is_cuda and x.is_contiguous() and x.dim() == 2
    out = torch.empty_like(x) if out is None else out
    with torch.cuda.device(x.device):
        L.check(L.load().dfft_fft1d_rows(x.data_ptr(), out.data_ptr(), x.shape[1], x.shape[0], _dtype_code(x), direction,
                                         None), "dfft_fft1d_rows")
        torch.cuda.synchronize()
    return out


def fft1d_cols(x, direction: int = FORWARD, out=None):
    """Length-n FFT down the columns of every (n, width) matrix of a (batch, n, width) complex device tensor."""
    import torch
    assert x.is_cuda and x.is_contiguous() and x.dim() == 3
    out = torch.empty_like(x) if out is None else out
    with torch.cuda.device(x.device):
        L.check(L.load().dfft_fft1d_cols(x.data_ptr(), out.data_ptr(), x.shape[1], x.shape[2], x.shape[0], _dtype_code(x),
                                         direction, None), "dfft_fft1d_cols")
        torch.cuda.synchronize()
    return out


def fft2d_batch(x, direction: int = FORWARD, out=None):
    """2D FFT of every (n1, n2) plane of a (batch, n1, n2) complex device tensor -- the t0 stage of a 3D plan as a call of its own
    (dfft_fft2d_batch; templateFFT's FFTDim = 2 application, templateFFT.cpp:5767).  out=x transforms in place."""
    import torch
    assert x.is_cuda and x.is_contiguous() and x.dim() == 3
    out = torch.empty_like(x) if out is None else out
    with torch.cuda.device(x.device):
        L.check(L.load().dfft_fft2d_batch(x.data_ptr(), out.data_ptr(), x.shape[1], x.shape[2], x.shape[0], _dtype_code(x), direction,
                                          None), "dfft_fft2d_batch")
        torch.cuda.synchronize()
    return out
