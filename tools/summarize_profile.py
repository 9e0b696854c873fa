"""Condense the rocprofv3 output of tools/profile_r01.sh (gpurun_out/prof_r01/) into the small files kept under profiles/:
  profiles/r01/bench_512_fp64_P1_kernel_stats.csv          rocprofv3 --stats rows of the default bench run (library kernels
                                                           with short names first, the rest as they are)
  profiles/r01/bench_512_fp64_P1_nochunk_kernel_stats.csv  same with DFFT_CHUNK_MB=0 (whole-slab Z / Y launches)
  profiles/r01/bench_512_fp64_P1_pmc_summary.csv           FETCH_SIZE / WRITE_SIZE per kernel (separate passes)
  profiles/hbm_traffic.json                                HBM bytes per X-pass launch for bench.py's roofline.traffic
FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: on gfx950 it reports half of the bytes of a wide coalesced streaming
read -- 128-byte requests tallied at 64 B); both counters are reported in KB and converted with x1024.  They sit on the L2's
fabric side, so Infinity-Cache hits are included."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "gpurun_out" / "prof_r01"
DST = ROOT / "profiles" / "r01"


def short(name: str) -> str:
    m = re.search(r"fft_tiles_kernel<(.*?), dfft::Plan<(\d+), (\d+)[^>]*>, (\d+), (\d+), (-?\d+), (true|false), dfft::(\w+)>", name)
    if not m:
        m2 = re.search(r"fft_generic_kernel<(.*?), (-?\d+)>", name)
        if m2:
            return f"fft_generic_kernel dir={m2.group(2)}"
        return name if len(name) <= 100 else name[:97] + "..."  # framework kernels of the harness (input generation, checks)
    ty = "cpair" if "cpair" in m.group(1) else ("f64" if "double" in m.group(1) else "f32")
    return (f"fft_tiles_kernel {ty} N={m.group(2)} E={m.group(3)} CB={m.group(4)} G={m.group(5)} dir={m.group(6)} "
            f"general={m.group(7)} tune={m.group(8)}")


def stats(run: str, out: str):
    files = glob.glob(str(SRC / run / "**" / "*kernel_stats.csv"), recursive=True)
    if not files:
        print("no kernel stats for", run)
        return
    rows = list(csv.DictReader(open(max(files, key=lambda f: Path(f).stat().st_size))))
    rows.sort(key=lambda r: ("dfft::" not in r["Name"], -float(r["TotalDurationNs"])))
    with open(DST / out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"],
                        r["StdDev"]])
    print("wrote", DST / out)


def pmc():
    out = []
    per = {}
    for run in ("pmc_fetch_default", "pmc_write_default", "pmc_fetch_nochunk", "pmc_write_nochunk"):
        files = glob.glob(str(SRC / run / "**" / "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        acc = defaultdict(list)
        for r in csv.DictReader(open(max(files, key=lambda f: Path(f).stat().st_size))):
            if "dfft::" not in r["Kernel_Name"] and "copyBuffer" not in r["Kernel_Name"]:
                continue  # only the library's kernels (and the runtime's copy kernel as a calibration point)
            key = (short(r["Kernel_Name"]), r["Counter_Name"], r.get("Grid_Size", ""), r.get("VGPR_Count", ""),
                   r.get("Accum_VGPR_Count", ""), r.get("LDS_Block_Size", ""), r.get("Scratch_Size", ""))
            acc[key].append(float(r["Counter_Value"]))
        for key, vals in acc.items():
            out.append([run, *key, len(vals), round(sum(vals) / len(vals), 1)])
            per[(run, key[0], key[1])] = max(per.get((run, key[0], key[1]), 0.0), sum(vals) / len(vals))
    if not out:
        print("no PMC runs found")
        return
    with open(DST / "bench_512_fp64_P1_pmc_summary.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["run", "kernel", "counter", "grid", "vgpr", "agpr", "lds", "scratch", "dispatches", "avg_value_KB"])
        w.writerows(out)
    xk = [k for k in per if "TuneTransposedStore" in k[1] and "N=512" in k[1] and "dir=1" in k[1]]
    fetch = max([per[k] for k in xk if k[0] == "pmc_fetch_default"], default=None)
    write = max([per[k] for k in xk if k[0] == "pmc_write_default"], default=None)
    if fetch and write:
        traffic = (2 * fetch + write) * 1024
        j = {"512x512x512_fp64_P1": {
            "source": "profiles/r01/bench_512_fp64_P1_pmc_summary.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                      "FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section; KB -> bytes x1024)",
            "fft_cols X(+transpose)": {"FETCH_SIZE_KB": round(fetch, 1), "WRITE_SIZE_KB": round(write, 1),
                                        "hbm_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": 2 * 16 * 512 ** 3}}}
        (ROOT / "profiles" / "hbm_traffic.json").write_text(json.dumps(j, indent=1) + "\n")
        print("X pass: traffic", traffic, "= %.4f x algorithmic" % (traffic / (2 * 16 * 512 ** 3)))


if __name__ == "__main__":
    DST.mkdir(parents=True, exist_ok=True)
    stats("trace_default", "bench_512_fp64_P1_kernel_stats.csv")
    stats("trace_nochunk", "bench_512_fp64_P1_nochunk_kernel_stats.csv")
    pmc()
