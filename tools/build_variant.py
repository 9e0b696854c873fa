"""Developer helper: a second build of the library with extra compile-time switches on the FFT kernel units, dfft_zy.hip and dfft_plan.cpp, for A/B runs on
the GPU box (DFFT_LIB=<path> selects it in the Python harness).   python tools/build_variant.py <name> -DSWITCH=value ...
-> distributedfft_amd/lib/libdfft_variant_<name>.so (linked against the HIP / RCCL runtime bundled with PyTorch)."""
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from distributedfft_amd import build as B  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
B.build()
obj = B.LIBDIR / f"obj_variant_{name}"
obj.mkdir(parents=True, exist_ok=True)
units = [(B.CSRC / "dfft_fft_inst.hip", obj / f"dfft_fft_inst_{g}.o", [f"-DDFFT_INST_GROUP={g}"] + flags) for g in range(B.NUM_INST_GROUPS)]
units.append((B.CSRC / "dfft_zy.hip", obj / "dfft_zy.o", flags))  # the one-launch YZ stage has build-time switches of its own
units.append((B.CSRC / "dfft_plan.cpp", obj / "dfft_plan.o", ["-x", "hip"] + flags))  # ... some of which the plan has to know (DFFT_ZY_ROW_PITCH)
with ThreadPoolExecutor(max_workers=8) as ex:
    list(ex.map(lambda u: B._run([B.HIPCC] + B.COMMON + u[2] + ["-c", str(u[0]), "-o", str(u[1])]), units))
others = [str(B.OBJ / f"{n}.o") for n in ("dfft_kernels", "dfft_generic", "dfft_long", "dfft_exchange", "dfft_bootstrap", "dfft_alloc", "dfft_trace")]
tl = B._torch_lib_dir()
out = B.LIBDIR / f"libdfft_variant_{name}.so"
B._run(["g++", "-shared", "-fPIC", "-o", str(out)] + [str(u[1]) for u in units] + others +
       ["-L" + str(tl), "-l:libamdhip64.so", "-l:librccl.so", "-lpthread", "-Wl,-rpath," + str(tl)])
print(out)
