#!/bin/bash
# round 6, call 30: rotation of the exchange rows in cache lines per plane (default 3), per-rank local work of configs 5 / 4 / 3
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
L=$OUT/rot_lines_sweep.log
: > $L
for rl in 1 2 3 5 7 9 11 13 17; do
  echo "## DFFT_ROT_LINES=$rl" >> $L
  DFFT_ROT_LINES=$rl python tools/local_by_P.py 2048x2048x1024 fp32 2 8 serial 2>&1 | grep "rot=1" >> $L
  DFFT_ROT_LINES=$rl python tools/local_by_P.py 1024x768x512 fp64 2 8 serial 2>&1 | grep "rot=1" >> $L
  DFFT_ROT_LINES=$rl python tools/local_by_P.py 512x512x512 fp64 2 4 serial 2>&1 | grep "rot=1" >> $L
done
cat $L
