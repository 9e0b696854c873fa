#!/bin/bash
# round 6, call 40: where the BACKWARD plans of the BASELINE shapes stand (per-rank local work, exchange off), next to the forward ones
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
L=$OUT/backward_status.log
: > $L
SPECS="512x512x512:fp64:1 512x512x512:fp64:4 1024x768x512:fp64:8 1024x768x512:fp64:1 2048x2048x1024:fp32:8 1024x1024x1024:fp32:1 2048x1024x512:fp64:1"
for rep in 1 2; do
  echo "## forward" >> $L
  timeout 600 python tools/lib_ab.py $SPECS 2>&1 | grep "sha" | cut -c1-330 >> $L
  echo "## backward" >> $L
  DFFT_AB_DIR=-1 timeout 600 python tools/lib_ab.py $SPECS 2>&1 | grep "sha" | cut -c1-330 >> $L
done
cat $L
