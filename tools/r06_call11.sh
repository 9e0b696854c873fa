#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
RECORDS_SKIP="" bash tools/records.sh r06 > $OUT/records.log 2>&1
bash tools/long_axis_pmc.sh $OUT/long_axis_pmc_final > $OUT/long_axis_pmc_final.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
