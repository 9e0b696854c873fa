#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
L=distributedfft_amd/lib/zy_litmus
{ timeout 300 $L 4000 0; timeout 300 $L 500 1; timeout 300 $L 500 2; timeout 300 $L 2000 3 64; } > $OUT/zy_litmus_short.log 2>&1; cat $OUT/zy_litmus_short.log
timeout 900 python -m pytest tests/test_gpu_litmus.py -q 2>&1 | tail -3
{ timeout 900 $L 400000 0; timeout 900 $L 400000 0 64; timeout 600 $L 100000 3 64; timeout 300 $L 20000 1; timeout 300 $L 20000 2; } > $OUT/zy_litmus_long.log 2>&1; cat $OUT/zy_litmus_long.log
