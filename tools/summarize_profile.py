"""Condense the rocprofv3 output of tools/profile_bench.sh (gpurun_out/prof_<round>/) into the small files kept under profiles/:
  profiles/<round>/bench_512_fp64_P1_kernel_stats.csv          rocprofv3 --stats rows of the default bench run (library kernels
                                                               with short names first, the rest as they are)
  profiles/<round>/bench_512_fp64_P1_nochunk_kernel_stats.csv  same with DFFT_CHUNK_MB=0 (whole-slab Z / Y launches)
  profiles/<round>/bench_512_fp64_P1_pmc_summary.csv           FETCH_SIZE / WRITE_SIZE per kernel (separate passes)
  profiles/hbm_traffic.json                                    fabric bytes per X-pass launch and per t0 stage for bench.py's
                                                               roofline.traffic, with the sha256 of the profiled library
usage: python tools/summarize_profile.py [raw dir, default gpurun_out/prof_<round>] [round, default r03]
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
ROUND = sys.argv[2] if len(sys.argv) > 2 else "r03"
SRC = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "gpurun_out" / f"prof_{ROUND}"
if not SRC.is_absolute():
    SRC = ROOT / SRC
DST = ROOT / "profiles" / ROUND


def short(name: str) -> str:
    m = re.search(r"fft_tiles_kernel<(.*?), dfft::Plan<(\d+), (\d+)[^>]*>, (\d+), (\d+), (-?\d+), (true|false), dfft::(\w+)>", name)
    mz = re.search(r"zy_chunk_kernel<dfft::Plan<(\d+),[^>]*>, dfft::Plan<(\d+),[^>]*>, (-?\d+)(?:, (true|false))?(?:, (true|false))?(?:, (-?\d+))?(?:, (true|false))?>", name)
    if mz:  # t0 as one persistent launch (dfft_zy.hip)
        return (f"zy_chunk_kernel f64 NZ={mz.group(1)} NY={mz.group(2)} dir={mz.group(3)}{' packed' if mz.group(4) == 'true' else ''}"
                f"{' lazy-publish' if mz.group(5) == 'true' else ''}{' inverse-rows-first' if mz.group(6) and mz.group(6) != mz.group(3) else ''}"
                f"{' all-parts' if mz.group(7) == 'true' else ''} (one-launch YZ stage)")
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
    rows = list(csv.DictReader(open(max(files, key=lambda f: Path(f).stat().st_mtime))))
    rows.sort(key=lambda r: ("dfft::" not in r["Name"], -float(r["TotalDurationNs"])))
    with open(DST / out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"],
                        r["StdDev"]])
    print("wrote", DST / out)


def timed_region(run: str, out: str, steps: int = 20):
    """The library's kernels over the LAST `steps` forward executes of the trace (bench.py's timed region).  The plain --stats
    averages above also contain the warm-up executes and dfft_plan_tune's probe launches of the X-pass kernel (7 per candidate
    buffer, most of them on slow placements), so the X pass reads slower there than it runs in the pipeline."""
    files = glob.glob(str(SRC / run / "**" / "*kernel_trace.csv"), recursive=True)
    if not files:
        print("no kernel trace for", run)
        return
    rows = [r for r in csv.DictReader(open(max(files, key=lambda f: Path(f).stat().st_mtime))) if "dfft::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [short(r["Kernel_Name"]) for r in rows]
    # bench.py ends with one backward execute (round-trip check): cut at the last forward X pass
    last = max(i for i, n in enumerate(names) if "TuneTransposedStore" in n and "dir=1" in n)
    xs = [i for i in range(last + 1) if "TuneTransposedStore" in names[i] and "dir=1" in names[i]]
    first = xs[-steps - 1] + 1 if len(xs) > steps else 0  # the region starts behind the X pass of the execute before it
    per = defaultdict(list)
    for i in range(first, last + 1):
        per[names[i]].append(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]))
    with open(DST / out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel (last %d forward executes of the trace = the timed region)" % steps, "Calls", "AverageNs", "MinNs", "MaxNs"])
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            if v:
                w.writerow([k, len(v), round(sum(v) / len(v), 1), min(v), max(v)])
    print("wrote", DST / out)


def pmc():
    out = []
    per = {}
    for run in ("pmc_fetch_default", "pmc_write_default", "pmc_fetch_nochunk", "pmc_write_nochunk"):
        files = glob.glob(str(SRC / run / "**" / "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        acc = defaultdict(list)
        for r in csv.DictReader(open(max(files, key=lambda f: Path(f).stat().st_mtime))):
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
    # t0 of one execute = every chunk launch of the Z-row and the Y-column kernel: dispatches per execute x mean per dispatch
    def t0_total(run):
        tot, nexec = 0.0, None
        xn = max([r[-2] for r in out if r[0] == run and "TuneTransposedStore" in r[1]], default=0)  # X launches = executes
        for r in out:
            zy = r[1].startswith("zy_chunk_kernel") and "dir=1" in r[1] and "inverse-rows-first" not in r[1]
            if r[0] != run or ((("N=512" not in r[1]) or ("dir=1" not in r[1]) or ("TuneTransposedStore" in r[1])) and not zy):
                continue
            if zy:
                tot += r[-1]            # one launch per execute: the mean per dispatch IS the stage's traffic
            elif xn:
                tot += r[-1] * r[-2] / xn
        return tot if xn else None
    if fetch and write:
        traffic = (2 * fetch + write) * 1024
        sha = None
        try:
            sha = (SRC / "library_sha256.txt").read_text().strip()
        except Exception:
            pass
        src = (f"profiles/{ROUND}/bench_512_fp64_P1_pmc_summary.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
               "FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section; KB -> bytes x1024; fabric-side counters, Infinity-Cache "
               "hits included)")
        ent = {"source": src, "library_sha256": sha,
               "fft_cols X(+transpose)": {"FETCH_SIZE_KB": round(fetch, 1), "WRITE_SIZE_KB": round(write, 1),
                                          "hbm_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": 2 * 16 * 512 ** 3}}
        f0, w0 = t0_total("pmc_fetch_default"), t0_total("pmc_write_default")
        if f0 and w0:
            ent["t0 chunk kernels (Z rows + Y columns)"] = {
                "FETCH_SIZE_KB": round(f0, 1), "WRITE_SIZE_KB": round(w0, 1), "hbm_bytes_per_launch": (2 * f0 + w0) * 1024,
                "algorithmic_bytes_per_launch": 2 * 16 * 512 ** 3,
                "note": "the whole t0 stage of one execute (one persistent launch, or the sum over its chunk launches); the Z -> Y "
                        "intermediate crosses the fabric twice (written to and read back from the Infinity Cache), hence ~2x the "
                        "algorithmic bytes of the stage"}
        (ROOT / "profiles" / "hbm_traffic.json").write_text(json.dumps({"512x512x512_fp64_P1": ent}, indent=1) + "\n")
        print("X pass: traffic", traffic, "= %.4f x algorithmic" % (traffic / (2 * 16 * 512 ** 3)))
        if f0 and w0:
            print("t0    : traffic", (2 * f0 + w0) * 1024, "= %.4f x algorithmic" % ((2 * f0 + w0) * 1024 / (2 * 16 * 512 ** 3)))


if __name__ == "__main__":
    DST.mkdir(parents=True, exist_ok=True)
    stats("trace_default", "bench_512_fp64_P1_kernel_stats.csv")
    timed_region("trace_default", "bench_512_fp64_P1_timed_region.csv")
    stats("trace_nochunk", "bench_512_fp64_P1_nochunk_kernel_stats.csv")
    pmc()
