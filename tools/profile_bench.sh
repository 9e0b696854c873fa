#!/bin/bash
# Profiling recipe of the graded bench line (run on the GPU box through gpurun):  tools/profile_bench.sh [round, default r03]
# Kernel trace + stats and the HBM PMC counters are collected in SEPARATE passes (gpurun refuses --pmc combined with runtime
# traces; FETCH_SIZE and WRITE_SIZE do not fit in one pass anyway: MI355X_MICROARCH.md "rocprofv3 PMC slots").  The raw
# output stays under gpurun_out/prof_<round>/ (scratch); tools/summarize_profile.py condenses it into profiles/<round>/ and
# profiles/hbm_traffic.json, stamped with the sha256 of the library that was profiled.
ROUND=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/prof_$ROUND; mkdir -p $OUT
sha256sum $R/distributedfft_amd/lib/libdfft_mi355x_pt.so | cut -d' ' -f1 > $OUT/library_sha256.txt
BENCH="python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -- $BENCH > $OUT/trace_default.log 2>&1
# PROFILE_SKIP_NOCHUNK=1: only the default (chunked) configuration -- three passes instead of six
[ -z "$PROFILE_SKIP_NOCHUNK" ] && DFFT_CHUNK_MB=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_nochunk -- $BENCH > $OUT/trace_nochunk.log 2>&1
BENCH2="python $R/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_default -- $BENCH2 > $OUT/pmc_fetch_default.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_default -- $BENCH2 > $OUT/pmc_write_default.log 2>&1
[ -z "$PROFILE_SKIP_NOCHUNK" ] && DFFT_CHUNK_MB=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_nochunk -- $BENCH2 > $OUT/pmc_fetch_nochunk.log 2>&1
[ -z "$PROFILE_SKIP_NOCHUNK" ] && DFFT_CHUNK_MB=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_nochunk -- $BENCH2 > $OUT/pmc_write_nochunk.log 2>&1
# keep only what is small enough to travel back
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*.db" -delete
du -sh $OUT
cd $R && python tools/summarize_profile.py gpurun_out/prof_$ROUND $ROUND
