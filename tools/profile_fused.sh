#!/bin/bash
# PMC passes (separate runs, as MI355X_MICROARCH.md prescribes) over a few variants of tools/fused_t0.
# usage: [BIN=zy_stream] tools/profile_fused.sh <outdir> "<variant substring>" ...   (BIN: tools/bin/<name>, default fused_t0)
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/$1; shift
mkdir -p $OUT
for v in "$@"; do
  tag=$(echo "$v" | tr -c 'A-Za-z0-9' '_')
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    ptag=$(echo "$pmc" | tr -c 'A-Za-z0-9' '_')
    timeout 120 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/raw_${tag}_${ptag} -o p -- $ROOT/tools/bin/${BIN:-fused_t0} 1 "$v" > $OUT/log_${tag}_${ptag}.txt 2>&1
  done
done
# condense: per kernel name, per counter: mean over dispatches
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
rows = []
for f in sorted(glob.glob(out + "/raw_*/**/*counter_collection.csv", recursive=True)):
    tag = f.split("/raw_")[1].split("/")[0]
    acc = collections.defaultdict(list)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            acc[(r["Kernel_Name"][:90], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        if "fused_yz" in k or "zy_stream" in k or "fft_tiles" in k:
            rows.append((tag, k, c, len(v), sum(v) / len(v)))
with open(out + "/pmc_summary.csv", "w") as fh:
    fh.write("run,kernel,counter,dispatches,mean_value\n")
    for r in rows:
        fh.write("%s,\"%s\",%s,%d,%.1f\n" % r)
print(open(out + "/pmc_summary.csv").read())
PY
rm -rf $OUT/raw_*
