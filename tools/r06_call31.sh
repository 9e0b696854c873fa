#!/bin/bash
# round 6, call 31: rotation of the exchange rows, config 5's rank at P = 8 (and P = 4): 2 / 3 / 4 / 6 lines per plane, four plans per point, two processes
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
L=$OUT/rot_lines_c5.log
: > $L
for rep in 1 2; do for rl in 3 2 4 6; do
  echo "## DFFT_ROT_LINES=$rl" >> $L
  DFFT_ROT_LINES=$rl python tools/local_by_P.py 2048x2048x1024 fp32 4 8 2>&1 | grep "rot=1" >> $L
done; done
for rl in 3 2 4; do
  echo "## DFFT_ROT_LINES=$rl" >> $L
  DFFT_ROT_LINES=$rl python tools/local_by_P.py 2048x2048x1024 fp32 3 4 serial 2>&1 | grep "rot=1" >> $L
  DFFT_ROT_LINES=$rl python tools/local_by_P.py 1024x1024x1024 fp32 3 4 serial 2>&1 | grep "rot=1" >> $L
  DFFT_ROT_LINES=$rl python tools/local_by_P.py 2048x1024x512 fp64 3 4 serial 2>&1 | grep "rot=1" >> $L
done
cat $L
