"""Condense rocprofv3 counter_collection.csv files of a probe run: per library kernel and per consecutive group of `group`
dispatches (tools/placement_pmc.py: one group = one plan) the mean duration and the mean of every collected counter.
usage: pmc_summary.py <dir with pass_*/ subdirs> <group size> [kernel substring, default TuneTransposedStore]"""
import csv
import glob
import re
import sys
from collections import OrderedDict, defaultdict

src, group = sys.argv[1], int(sys.argv[2])
want = sys.argv[3] if len(sys.argv) > 3 else "TuneTransposedStore"
csv.field_size_limit(1 << 30)
for f in sorted(glob.glob(src + "/pass_*/**/*counter_collection.csv", recursive=True)):
    per = OrderedDict()  # dispatch id -> {counter: value, "_dur": ns}
    for r in csv.DictReader(open(f)):
        if want not in r["Kernel_Name"]:
            continue
        d = per.setdefault(int(r["Dispatch_Id"]), {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["_dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    ids = sorted(per)
    print("==", f.split("/pass_")[1].split("/")[0], "dispatches of", want, ":", len(ids))
    names = sorted({k for d in per.values() for k in d})
    print("group," + ",".join(names))
    for g in range(0, len(ids), group):
        chunk = [per[i] for i in ids[g:g + group]]
        tail = chunk[len(chunk) // 2:] if group < 100000 else chunk  # per-plan groups: second half = steady state
        print("%d," % (g // group) + ",".join("%.6g" % (sum(d.get(n, 0.0) for d in tail) / len(tail)) for n in names))
