#!/bin/bash
# round 6, call 26: the round's records for the final library (tools/records.sh) + smoke
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
RECORDS_SKIP="" bash tools/records.sh r06 > $OUT/records.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
tail -3 $OUT/pytest_gpu_full.log; cat $OUT/bench_driver_args.json
