"""Limiter table from the counter passes of tools/long_axis_pmc.sh (rocprofv3 --pmc over tools/long_axis_probe.py): one row per library
kernel (and launch geometry), the counters of all passes joined on the kernel, derived columns:
  us            mean launch duration under the counters (of the sq pass)
  rd/wr GB      fabric-side bytes per launch: FETCH_SIZE x 2 (MI355X_MICROARCH.md, HBM section) and WRITE_SIZE, KB -> x1024
  TB/s          (rd + wr) / us
  rd_lat wr_lat mean fabric latency in L2 cycles: TCC_EA0_RDREQ_LEVEL / RDREQ, WRREQ_LEVEL / WRREQ
  l2hit         TCC_HIT / (TCC_HIT + TCC_MISS)
  wait          SQ_WAIT_ANY / SQ_WAVE_CYCLES            (fraction of resident wave time spent waiting on a counter)
  wait_inst     SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES       (waiting for an instruction issue slot)
  active        SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  lds_wait      SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES
  lds/wave      SQ_INSTS_LDS / SQ_WAVES
  bank%         SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  tcp_stall     TCP_PENDING_STALL_CYCLES / (SQ_BUSY_CYCLES)   (L1 stalled on outstanding requests, per busy cycle of the SQ pass)
usage: long_axis_pmc_table.py <dir with pass_*/ subdirs> [min dispatches]"""
import csv
import glob
import re
import sys
from collections import defaultdict

src = sys.argv[1]
csv.field_size_limit(1 << 30)


def short(name: str) -> str:
    ty = lambda s: "pair" if "cpair" in s else ("f64" if "double" in s else "f32")  # noqa: E731
    m = re.search(r"fft_tiles_kernel<(.*?), dfft::Plan<(\d+), (\d+)[^>]*>, (\d+), (\d+), (-?\d+), (true|false), (.*)>\(", name)
    if m:
        tune = re.sub(r"dfft::", "", m.group(8))
        return f"tiles {ty(m.group(1))} N={m.group(2)} E={m.group(3)} CB={m.group(4)} dir={m.group(6)} {tune}"
    m = re.search(r"zy_chunk_kernel<dfft::Plan<(\d+),[^>]*>, dfft::Plan<(\d+),[^>]*>, (-?\d+), (true|false), (true|false)", name)
    if m:
        return f"zy_chunk NZ={m.group(1)} NY={m.group(2)} dir={m.group(3)}{' packed' if m.group(4) == 'true' else ''}{' lazy' if m.group(5) == 'true' else ''}"
    m = re.search(r"(fft_dual_tiles_kernel|fft_dif2_tiles_kernel|fft_tload_tiles_kernel)<(.*?), dfft::Plan<(\d+), (\d+)[^>]*>, (\d+), (-?\d+)(.*)>\(", name)
    if m:
        n = int(m.group(3)) * (2 if "dif2" in m.group(1) else 1)
        return f"{m.group(1)[4:-13]} {ty(m.group(2))} N={n} E={m.group(4)} CB={m.group(5)} dir={m.group(6)}{m.group(7).replace(' ', '')}"
    return None


acc = defaultdict(lambda: defaultdict(list))  # (kernel, grid, wg) -> counter -> [values per dispatch]
dur = defaultdict(list)
res = {}
for f in sorted(glob.glob(src + "/pass_*/**/*counter_collection.csv", recursive=True)):
    per = {}
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k is None:
            continue
        key = (k, r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))
        d = per.setdefault((key, int(r["Dispatch_Id"])), {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        res[key] = (r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""), r.get("LDS_Block_Size", ""), r.get("Scratch_Size", ""))
    for (key, _), d in per.items():
        for n, v in d.items():
            if n == "_us":
                if "pass_sq" in f:
                    dur[key].append(v)
            else:
                acc[key][n].append(v)
mind = int(sys.argv[2]) if len(sys.argv) > 2 else 2


def mean(key, n):
    v = acc[key].get(n)
    if not v:
        return None
    v = v[len(v) // 3:]  # skip the first launches (cold caches, first-touch)
    return sum(v) / len(v)


def ratio(a, b, scale=1.0):
    return None if a is None or not b else scale * a / b


def fmt(v, f="%.2f"):
    return "-" if v is None else f % v


print("| kernel | grid x wg | vgpr+agpr / lds KiB | launches | us | rd GB | wr GB | TB/s | rd_lat | wr_lat | l2hit | wait | wait_inst | active | lds_wait | lds/wave | bank% | tcp_stall |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for key in sorted(dur, key=lambda k: (k[0], k[1])):
    if len(dur[key]) < mind:
        continue
    d = dur[key][len(dur[key]) // 3:]
    us = sum(d) / len(d)
    m = lambda n: mean(key, n)  # noqa: E731
    rd = None if m("FETCH_SIZE") is None else 2 * m("FETCH_SIZE") * 1024 / 1e9
    wr = None if m("WRITE_SIZE") is None else m("WRITE_SIZE") * 1024 / 1e9
    tb = None if rd is None or wr is None else (rd + wr) * 1e9 / (us * 1e-6) / 1e12
    wc = m("SQ_WAVE_CYCLES")
    vg, ag, lds, _ = res[key]
    print("| " + " | ".join([
        key[0], f"{key[1]} x {key[2]}", f"{vg}+{ag} / {int(lds or 0) // 1024}", str(len(dur[key])), "%.1f" % us, fmt(rd, "%.3f"), fmt(wr, "%.3f"), fmt(tb),
        fmt(ratio(m("TCC_EA0_RDREQ_LEVEL_sum"), m("TCC_EA0_RDREQ_sum")), "%.0f"), fmt(ratio(m("TCC_EA0_WRREQ_LEVEL_sum"), m("TCC_EA0_WRREQ_sum")), "%.0f"),
        fmt(ratio(m("TCC_HIT_sum"), (m("TCC_HIT_sum") or 0) + (m("TCC_MISS_sum") or 0))),
        fmt(ratio(m("SQ_WAIT_ANY"), wc)), fmt(ratio(m("SQ_WAIT_INST_ANY"), wc)), fmt(ratio(m("SQ_ACTIVE_INST_ANY"), wc)), fmt(ratio(m("SQ_WAIT_INST_LDS"), wc)),
        fmt(ratio(m("SQ_INSTS_LDS"), m("SQ_WAVES")), "%.0f"), fmt(ratio(m("SQ_LDS_BANK_CONFLICT"), m("SQ_LDS_IDX_ACTIVE"), 100.0), "%.1f"),
        fmt(ratio(m("TCP_PENDING_STALL_CYCLES_sum"), m("SQ_BUSY_CYCLES")))]) + " |")
