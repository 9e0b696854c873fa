"""Build libdfft_mi355x.so and the distFFTOpt driver for gfx950 with hipcc (cross-compiles without a GPU).

Artefacts stay in-tree (distributedfft_amd/lib/) so they travel with the repo snapshot to the GPU box; they are
git-ignored.  Usage:  python -m distributedfft_amd.build [--force] [--jobs N]
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
OBJ = PKG / "lib" / "obj"
LIBDIR = PKG / "lib"
INCLUDE = ROOT / "include"
ROCM = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))
HIPCC = str(ROCM / "bin" / "hipcc")
ARCH = "gfx950"
NUM_INST_GROUPS = 12  # keep in sync with csrc/dfft_plans.h

# --offload-compress: the gfx950 code objects are stored zstd-compressed and unpacked by the HIP runtime when the library is loaded
# (65 MB -> 18.5 MB per .so; process start and the benchmark unchanged, profiles/r05/experiments/compress_smoke.log)
COMMON = ["--offload-arch=" + ARCH, "--offload-compress", "-O3", "-std=c++17", "-fPIC", "-I" + str(INCLUDE), "-I" + str(CSRC),
          "-Wno-unused-result"]

LIB_NAME = "libdfft_mi355x.so"       # links /opt/rocm (standalone C++ applications, distFFTOpt)
LIB_NAME_PT = "libdfft_mi355x_pt.so"  # links the runtime bundled with PyTorch (Python hosts)


def _torch_lib_dir():
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return None
        d = Path(spec.origin).parent / "lib"
        return d if (d / "libamdhip64.so").exists() and (d / "librccl.so").exists() else None
    except Exception:
        return None


def _newer(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
    return r


def _headers():
    return sorted(list(CSRC.glob("*.h")) + list(INCLUDE.glob("*.h")) + list(INCLUDE.glob("*/*.h")))


def _deps(src: Path, seen=None):
    """src and every project header it includes, transitively (quoted includes only): a change to a host-side header
    must not recompile the twelve kernel instantiation groups."""
    import re
    seen = set() if seen is None else seen
    src = src.resolve()
    if src in seen or not src.exists():
        return seen
    seen.add(src)
    for m in re.finditer(r'^\s*#\s*include\s+"([^"]+)"', src.read_text(errors="replace"), re.M):
        for base in (src.parent, CSRC, INCLUDE):
            cand = (base / m.group(1))
            if cand.exists():
                _deps(cand, seen)
                break
    return seen


def build(force: bool = False, jobs: int | None = None, verbose: bool = False) -> Path:
    OBJ.mkdir(parents=True, exist_ok=True)
    hdrs = _headers()
    units = []  # (source, object, extra flags)
    for g in range(NUM_INST_GROUPS):
        units.append((CSRC / "dfft_fft_inst.hip", OBJ / f"dfft_fft_inst_{g}.o", [f"-DDFFT_INST_GROUP={g}"]))
    units.append((CSRC / "dfft_kernels.hip", OBJ / "dfft_kernels.o", []))
    units.append((CSRC / "dfft_generic.hip", OBJ / "dfft_generic.o", []))
    units.append((CSRC / "dfft_long.hip", OBJ / "dfft_long.o", []))
    units.append((CSRC / "dfft_zy.hip", OBJ / "dfft_zy.o", []))
    for name in ("dfft_plan", "dfft_exchange", "dfft_bootstrap", "dfft_alloc", "dfft_trace"):
        units.append((CSRC / f"{name}.cpp", OBJ / f"{name}.o", ["-x", "hip"]))

    def compile_one(u):
        src, obj, extra = u
        if not force and not _newer(obj, _deps(src)):
            return False
        cmd = [HIPCC] + COMMON + extra + ["-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        _run(cmd)
        return True

    jobs = jobs or min(8, os.cpu_count() or 4)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        rebuilt = list(ex.map(compile_one, units))

    lib = LIBDIR / LIB_NAME
    objs = [str(u[1]) for u in units]
    if force or any(rebuilt) or _newer(lib, objs):
        _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", str(lib)] + objs +
             ["-L" + str(ROCM / "lib"), "-lrccl", "-Wl,-rpath," + str(ROCM / "lib")])

    # Same objects linked against the HIP/RCCL runtime that PyTorch bundles (torch/lib): a Python process that has
    # imported torch must not get a second HIP runtime from /opt/rocm next to torch's own (two runtimes in one
    # process = sporadic hipErrorUnknown and heap corruption at exit).  ctypes loads this variant (_lib.py).
    tl = _torch_lib_dir()
    if tl is not None:
        lib_pt = LIBDIR / LIB_NAME_PT
        if force or any(rebuilt) or _newer(lib_pt, objs):
            _run(["g++", "-shared", "-fPIC", "-o", str(lib_pt)] + objs +
                 ["-L" + str(tl), "-l:libamdhip64.so", "-l:librccl.so", "-lpthread", "-Wl,-rpath," + str(tl)])

    # driver: our clone, built against the MPI shim (no MPI installation needed)
    drv = LIBDIR / "distFFTOpt"
    drv_src = CSRC / "distFFTOpt.cpp"
    if force or _newer(drv, [drv_src, lib] + hdrs):
        _run([HIPCC] + COMMON + ["-x", "hip", "-I" + str(INCLUDE / "dfft_mpi_shim"), str(drv_src), "-x", "none",
                                 "-o", str(drv), "-L" + str(LIBDIR), "-ldfft_mi355x", "-lpthread",
                                 "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + str(ROCM / "lib")])
    # heFFTe-protocol benchmark front-end on top of the C-ABI (same CLI/report as the CPU baseline's speed3d_c2c)
    s3d = LIBDIR / "speed3d_c2c"
    s3d_src = CSRC / "speed3d_c2c_dfft.cpp"
    if force or _newer(s3d, [s3d_src, lib] + hdrs):
        _run([HIPCC] + COMMON + ["-x", "hip", str(s3d_src), "-x", "none", "-o", str(s3d), "-L" + str(LIBDIR),
                                 "-ldfft_mi355x", "-lpthread", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + str(ROCM / "lib")])
    # Test_1D / Test_2D: the reference's batched component benchmarks (templateFFT/batchTest) on the C-ABI
    bt_src = CSRC / "batch_test.cpp"
    for dim in (1, 2):
        exe = LIBDIR / f"Test_{dim}D"
        if force or _newer(exe, [bt_src, lib] + hdrs):
            _run([HIPCC] + COMMON + ["-x", "hip", f"-DBATCH_DIM={dim}", str(bt_src), "-x", "none", "-o", str(exe), "-L" + str(LIBDIR),
                                     "-ldfft_mi355x", "-lpthread", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + str(ROCM / "lib")])
    # message-passing litmus of the hand-offs the one-launch YZ stage relies on (tools/zy_litmus.hip; tests/test_gpu_litmus.py runs it)
    lit = LIBDIR / "zy_litmus"
    lit_src = ROOT / "tools" / "zy_litmus.hip"
    if lit_src.exists() and (force or _newer(lit, [lit_src])):
        _run([HIPCC, "--offload-arch=" + ARCH, "-O3", "-std=c++17", str(lit_src), "-o", str(lit)])
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    ap.add_argument("-v", "--verbose", action="store_true")
    a = ap.parse_args()
    lib = build(a.force, a.jobs, a.verbose)
    print(lib)


if __name__ == "__main__":
    sys.exit(main())
