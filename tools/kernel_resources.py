"""Register / scratch / LDS usage of the FFT kernels of one instantiation group (no GPU needed): compiles
csrc/dfft_fft_inst.hip for gfx950 with -save-temps and reads the .amdhsa_* directives of every fft_tiles_kernel / fft_dual_tiles_kernel.

  python tools/kernel_resources.py <group> [filter]        e.g.  python tools/kernel_resources.py 3 N=512
  python tools/kernel_resources.py zy                      the kernels of the one-launch YZ stage (csrc/dfft_zy.hip)
  python tools/kernel_resources.py all <out.txt> [jobs]    every group + zy into one inventory whose first line carries the sha256 of the
                                                           kernel sources (tests/test_host_logic.py checks that the committed inventory
                                                           belongs to the sources in the tree and that nothing up to 2048 points spills)

A kernel that spills (scratch > 0) or loses occupancy shows up here long before it shows up in a benchmark: the 16- and
24-point-per-thread column kernels sit within a few registers of the 256-VGPR budget of a 512-thread block."""
import os
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "distributedfft_amd" / "csrc"

TYPES = {"15HIP_vector_typeIdLj2EE": "f64", "15HIP_vector_typeIfLj2EE": "f32", "NS_5cpairE": "pair"}


def kernel_table(group: int):
    """[(tag, vgprs incl. AGPRs, scratch bytes, static LDS bytes)] for instantiation group `group`."""
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT / 'include'}", f"-I{CSRC}",
               f"-DDFFT_INST_GROUP={group}", "-c", str(CSRC / "dfft_fft_inst.hip"), "-o", "inst.o", "-save-temps"]
        cmd += os.environ.get("DFFT_KR_FLAGS", "").split()  # extra -D switches for A/B inventories
        r = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-3000:])
        asm = next(Path(tmp).glob("*gfx950.s")).read_text()
    rows = []
    for m in re.finditer(r"^(_ZN4dfft(?:16fft_tiles_kernel|21fft_dual_tiles_kernel|21fft_dif2_tiles_kernel|22fft_tload_tiles_kernel)\w+): ", asm, re.M):
        name = m.group(1)
        body = asm[m.start():asm.index(".end_amdhsa_kernel", m.start())]
        mm = re.match(r"_ZN4dfft16fft_tiles_kernelI(.*?)NS_4PlanILi(\d+)ELi(\d+)EJ.*?EEELi(\d+)ELi(\d+)ELi(n?1)ELb([01])ENS_\d+(\w+?)(?:ILi\d+EE)?EEEv", name)
        md = re.match(r"_ZN4dfft21fft_dual_tiles_kernelI(.*?)NS_4PlanILi(\d+)ELi(\d+)EJ.*?EEELi(\d+)ELi(n?1)ELb([01])E(?:Lb([01])E)?(?:Li(\d+)E)?EEv", name)
        if mm:
            ty = TYPES.get(mm.group(1), mm.group(1))
            tag = (f"{ty} N={mm.group(2)} E={mm.group(3)} CB={mm.group(4)} G={mm.group(5)} dir={'-1' if mm.group(6) == 'n1' else '1'} "
                   f"general={mm.group(7)} {mm.group(8)}")
        elif md:  # paired half-line tiles (transposing store of the 2048-point X pass)
            ty = TYPES.get(md.group(1), md.group(1))
            tag = (f"{ty} N={md.group(2)} E={md.group(3)} CB=2x{md.group(4)} G=1 dir={'-1' if md.group(5) == 'n1' else '1'} general=0 DualTiles"
                   f"{' rotated-rows' if md.group(7) == '1' else ''}{' x' + md.group(8) + ' per CU' if md.group(8) and md.group(8) != '1' else ''}")
        elif "fft_tload_tiles_kernel" in name:  # staged transposed load (inverse X pass)
            mt = re.match(r"_ZN4dfft22fft_tload_tiles_kernelI(.*?)NS_4PlanILi(\d+)ELi(\d+)EJ.*?EEELi(\d+)ELi(n?1)ELb([01])EEEv", name)
            if not mt:
                continue
            ty = TYPES.get(mt.group(1), mt.group(1))
            tag = (f"{ty} N={mt.group(2)} E={mt.group(3)} CB={mt.group(4)} G=1 dir={'-1' if mt.group(5) == 'n1' else '1'} general=0 "
                   f"TransposedLoad{' rotated-rows' if mt.group(6) == '1' else ''}")
        elif "fft_dif2_tiles_kernel" in name:  # DIF-split full-line tiles (non-transposing 2048-point column passes)
            mh = re.match(r"_ZN4dfft21fft_dif2_tiles_kernelI(.*?)NS_4PlanILi(\d+)ELi(\d+)EJ.*?EEELi(\d+)ELi(n?1)ELb([01])ELb([01])E", name)
            if not mh:
                continue
            ty = TYPES.get(mh.group(1), mh.group(1))
            tag = (f"{ty} N={2 * int(mh.group(2))} E={2 * int(mh.group(3))} CB={mh.group(4)} G=1 dir={'-1' if mh.group(5) == 'n1' else '1'} "
                   f"general=0 Dif2Tiles ntl={mh.group(6)} nts={mh.group(7)}")
        else:
            continue

        def field(key):
            f = re.search(rf"\.amdhsa_{key} (\d+)", body)
            return int(f.group(1)) if f else 0
        rows.append((tag, field("next_free_vgpr"), field("private_segment_fixed_size"), field("group_segment_fixed_size")))
    return rows


def zy_table():
    """[(tag, vgprs, scratch bytes, static LDS bytes)] of the one-launch YZ stage's kernels (csrc/dfft_zy.hip)."""
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT / 'include'}", f"-I{CSRC}",
               "-c", str(CSRC / "dfft_zy.hip"), "-o", "zy.o", "-save-temps"]
        cmd += os.environ.get("DFFT_KR_FLAGS", "").split()
        r = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-3000:])
        asm = next(Path(tmp).glob("*gfx950.s")).read_text()
    rows = []
    for m in re.finditer(r"^(_ZN4dfft\w*zy_chunk_kernel\w+): ", asm, re.M):
        name = m.group(1)
        body = asm[m.start():asm.index(".end_amdhsa_kernel", m.start())]
        lens = re.findall(r"ILi(\d+)ELi(?:8|12|24)E", name)  # Plan<N, 8 | 12 | 24, ...> of the Z and (when different) Y axis
        mm = re.search(r"Li(n?1)ELb([01])ELb([01])E(?:Li(n?1)E)?(?:Lb([01])E)?", name)
        if not lens or not mm:
            continue
        nz, ny = lens[0], lens[-1]

        def field(key):
            f = re.search(rf"\.amdhsa_{key} (\d+)", body)
            return int(f.group(1)) if f else 0
        tag = (f"f64 zy_chunk_kernel Z={nz} Y={ny} dir={'-1' if mm.group(1) == 'n1' else '1'} packed={mm.group(2)} "
               f"{'lazy' if mm.group(3) == '1' else 'eager'}{' inverse-rows-first' if mm.group(4) == 'n1' and mm.group(1) == '1' else ''}"
               f"{' all-parts (counts finished column units per exchange part)' if mm.group(5) == '1' else ''}")
        rows.append((tag, field("next_free_vgpr"), field("private_segment_fixed_size"), field("group_segment_fixed_size")))
    return rows


KERNEL_SOURCES = ("dfft_fft_impl.h", "dfft_butterfly.h", "dfft_plans.h", "dfft_fft_inst.hip", "dfft_zy.hip", "dfft_zy.h", "dfft_kernels.h")


def sources_sha256() -> str:
    """sha256 over the files every FFT kernel instantiation is generated from."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        h.update(name.encode())
        h.update((CSRC / name).read_bytes())
    return h.hexdigest()


def num_groups() -> int:
    return int(re.search(r"#define DFFT_NUM_INST_GROUPS (\d+)", (CSRC / "dfft_plans.h").read_text()).group(1))


def write_inventory(out: Path, jobs: int = 4):
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        tables = list(ex.map(kernel_table, range(num_groups()))) + [zy_table()]
    lines = [f"# kernel sources sha256 {sources_sha256()}  ({' '.join(KERNEL_SOURCES)}; tools/kernel_resources.py all)"]
    for rows in tables:
        for tag, vgpr, scratch, lds in rows:
            lines.append(f"{tag:<70} vgpr {vgpr:>3}  scratch {scratch:>4} B")
    out.write_text("\n".join(lines) + "\n")
    return len(lines) - 1


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "all":
        n = write_inventory(Path(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 4)
        print(f"{n} kernels -> {sys.argv[2]}")
        sys.exit(0)
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = zy_table() if len(sys.argv) > 1 and sys.argv[1] == "zy" else kernel_table(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
    for tag, vgpr, scratch, lds in rows:
        if flt in tag:
            print(f"{tag:<70} vgpr {vgpr:>3}  scratch {scratch:>4} B")
