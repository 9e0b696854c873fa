"""ctypes loader for libdfft_mi355x.so (the C-ABI declared in include/dfft.h).

There is deliberately no fallback: if the shared library is missing or does not load, importing callers get an
ImportError that says how to build it.  Nothing in this package computes an FFT on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH_SYSTEM = _PKG / "lib" / "libdfft_mi355x.so"   # linked against /opt/rocm: for C++ hosts (distFFTOpt)
LIB_PATH = _PKG / "lib" / "libdfft_mi355x_pt.so"       # same objects linked against PyTorch's bundled HIP/RCCL runtime
DRIVER_PATH = _PKG / "lib" / "distFFTOpt"

# constants mirrored from include/dfft.h
FORWARD, BACKWARD = 1, -1
ALLOC_HOST, ALLOC_DEV = 1, -1
F64, F32 = 0, 1
PLAN_DEFAULT, PLAN_UNFUSED, PLAN_INPUT_FROM_IN, PLAN_OVERLAP, PLAN_NATURAL = 0, 1, 2, 4, 8
EXEC_ASYNC, EXEC_SYNC_STAGES, EXEC_PRINT, EXEC_NO_TIMING = 0, 1, 2, 4
OK, EINVAL, EHIP, ERCCL, ENOGPU, ECOMM, EUNSUPPORTED = 0, -1, -2, -3, -4, -5, -6

_LL = C.c_longlong
_LLP = C.POINTER(C.c_longlong)
_VP = C.c_void_p

# name -> (restype, argtypes); every symbol include/dfft.h declares
SIGNATURES = {
    "dfft_version": (C.c_char_p, []),
    "dfft_last_error": (C.c_char_p, []),
    "dfft_device_count": (C.c_int, []),
    "dfft_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "dfft_length_supported": (C.c_int, [_LL]),
    "dfft_proper_device_count": (C.c_int, [_LLP, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dfft_local_count": (_LL, [_LLP, C.c_int, C.c_int]),
    "dfft_max_count": (_LL, [_LL, _LL, _LL, C.c_int, C.c_int]),
    "dfft_exchange_layout": (C.c_int, [_LL, _LL, _LL, C.c_int, C.c_int, C.c_int, _LLP, _LLP, _LLP, _LLP]),
    "dfft_exchange_part_layout": (C.c_int, [_LL, _LL, _LL, C.c_int, C.c_int, C.c_int, _LL, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.POINTER(C.c_int), _LLP, _LLP, _LLP, _LLP]),
    "dfft_local_size": (C.c_int, [_LL, _LL, _LL, C.c_int, C.c_int, _LLP, _LLP, _LLP, _LLP]),
    "dfft_comm_create_local": (C.c_int, [C.c_int, C.POINTER(_VP)]),
    "dfft_rccl_unique_id": (C.c_int, [C.c_char_p]),
    "dfft_comm_create_rccl": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(_VP)]),
    "dfft_comm_create_ipc": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_VP)]),
    "dfft_comm_info": (C.c_int, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dfft_comm_destroy": (C.c_int, [_VP]),
    "dfft_alloc": (_VP, [_LL, C.c_int, C.c_int]),
    "dfft_free": (C.c_int, [_VP, C.c_int]),
    "dfft_plan_create": (C.c_int, [C.POINTER(_VP), _LL, _LL, _LL, C.c_int, C.c_int, _VP, _VP, _VP, C.c_int, C.c_int, C.c_uint]),
    "dfft_plan_buffer1": (_VP, [_VP]),
    "dfft_plan_result": (_VP, [_VP]),
    "dfft_plan_stream": (_VP, [_VP]),
    "dfft_plan_workbuf": (_VP, [_VP, _LLP]),
    "dfft_fft2d_batch": (C.c_int, [_VP, _VP, _LL, _LL, _LL, C.c_int, C.c_int, _VP]),
    "dfft_execute": (C.c_int, [_VP, C.c_uint]),
    "dfft_plan_sync": (C.c_int, [_VP]),
    "dfft_plan_tune": (C.c_int, [_VP]),
    "dfft_plan_describe": (C.c_int, [_VP, C.c_char_p, C.c_int]),
    "dfft_plan_tune_report": (C.c_int, [_VP, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "dfft_plan_set_scale": (C.c_int, [_VP, C.c_double]),
    "dfft_stage_times": (C.c_int, [_VP, C.POINTER(C.c_double)]),
    "dfft_kernel_times": (C.c_int, [_VP, C.POINTER(C.c_double)]),
    "dfft_plan_destroy": (C.c_int, [_VP]),
    "dfft_fft1d_rows": (C.c_int, [_VP, _VP, _LL, _LL, C.c_int, C.c_int, _VP]),
    "dfft_fft1d_cols": (C.c_int, [_VP, _VP, _LL, _LL, _LL, C.c_int, C.c_int, _VP]),
    "dfft_scale": (C.c_int, [_VP, _LL, C.c_int, C.c_double, _VP]),
    "dfft_trim": (C.c_int, []),
    "dfft_boot_init": (C.c_int, []),
    "dfft_boot_rank": (C.c_int, []),
    "dfft_boot_size": (C.c_int, []),
    "dfft_boot_bcast": (C.c_int, [_VP, C.c_size_t, C.c_int]),
    "dfft_boot_barrier": (C.c_int, []),
    "dfft_boot_allreduce_max": (C.c_int, [C.POINTER(C.c_double), C.c_int]),
    "dfft_boot_finalize": (C.c_int, []),
    "dfft_trace_dump": (C.c_int, []),
}

_lib = None


class DfftError(RuntimeError):
    def __init__(self, code: int, where: str, msg: str):
        super().__init__(f"{where} failed with {code}: {msg}")
        self.code = code


def load() -> C.CDLL:
    """Load the native library (once).  Raises ImportError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("DFFT_LIB", str(LIB_PATH)))
    try:  # make sure the HIP runtime this variant binds to is torch's, already resident in the process
        import torch  # noqa: F401
    except ImportError:
        if "DFFT_LIB" not in os.environ:
            path = LIB_PATH_SYSTEM
    if not path.exists():
        raise ImportError(
            f"{path} not found: build the HIP extension first (python -m distributedfft_amd.build, or "
            f"__graft_entry__.build()).  distributedfft_amd has no CPU fallback.")
    try:
        lib = C.CDLL(str(path), mode=C.RTLD_GLOBAL)
    except OSError as e:  # missing ROCm runtime etc.
        raise ImportError(f"cannot load {path}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)  # AttributeError here means header and library disagree
        except AttributeError:
            if "DFFT_LIB" in os.environ:  # a developer's A/B build of an older source tree (tools/lib_ab.py): entry points added since are absent
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int, where: str) -> None:
    if code != OK:
        raise DfftError(code, where, load().dfft_last_error().decode("utf-8", "replace"))
