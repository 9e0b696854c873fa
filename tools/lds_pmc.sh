#!/bin/bash
# LDS pressure of the pipeline's kernels (GPU box): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE / SQ_BUSY_CYCLES per kernel, one
# counter pass, for the graded 512^3 fp64 bench and for a 2048-point X pass.  tools/lds_pmc.sh [round]
ROUND=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/$ROUND/lds_pmc; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/b512 -- \
    python $R/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/b512.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/x2048 -- \
    python $R/tools/xpass_ab.py > $OUT/x2048.log 2>&1
find $OUT -name "*.db" -delete
python - <<PY
import csv, glob, collections
for tag in ("b512", "x2048"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:110]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_BUSY_CYCLES": cnt[k] += 1
    print("==", tag)
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0))[:8]:
        act = v.get("SQ_LDS_IDX_ACTIVE", 0); bc = v.get("SQ_LDS_BANK_CONFLICT", 0)
        print(f"{k}\n   launches {cnt[k]}  lds_active {act:.3g}  bank_conflict {bc:.3g} ({100*bc/max(act,1):.1f} % of active)  inst_lds {v.get('SQ_ACTIVE_INST_LDS',0):.3g}  busy {v.get('SQ_BUSY_CYCLES',0):.3g}")
PY
