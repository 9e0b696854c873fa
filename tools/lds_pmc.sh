#!/bin/bash
# LDS pressure of the long-axis X-pass kernels (GPU box): bank conflicts / LDS activity / wave-state counters per kernel for the
# 1024- and 2048-point X passes in fp64 and on fp32 column pairs.   tools/lds_pmc.sh [outdir]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=${1:-$R/gpurun_out/r03/lds_pmc}; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/x -- \
    python $R/tools/xpass_variants.py 2048x256x512 1024x512x512 512x512x512 > $OUT/x.log 2>&1
find $OUT -name "*.db" -delete
python - <<PY
import csv, glob, collections, re
csv.field_size_limit(1 << 30)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(float)
for f in glob.glob("$OUT/x/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "dfft::" not in k or ("TuneTransposedStore" not in k and "dual" not in k):
            continue
        m = re.search(r"(fft_\w+)<(.*?), dfft::Plan<(\d+)", k)
        key = "%s %s N=%s" % (m.group(1), "pair" if "cpair" in m.group(2) else "f64", m.group(3)) if m else k[:80]
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            cnt[key] += 1
            dur[key] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
for k, v in sorted(agg.items()):
    n = max(cnt[k], 1); wc = v["SQ_WAVE_CYCLES"]
    print(f"{k:40s} launches {n:3d} avg {dur[k]/n:8.1f} us | LDS active {v['SQ_LDS_IDX_ACTIVE']/n:.3g} bank conflict {v['SQ_LDS_BANK_CONFLICT']/n:.3g} ({100*v['SQ_LDS_BANK_CONFLICT']/max(v['SQ_LDS_IDX_ACTIVE'],1):.1f} %)"
          f" | wait_any {v['SQ_WAIT_ANY']/wc:.2f} wait_inst {v['SQ_WAIT_INST_ANY']/wc:.2f} (lds {v['SQ_WAIT_INST_LDS']/wc:.2f}) active {v['SQ_ACTIVE_INST_ANY']/wc:.2f}")
PY
