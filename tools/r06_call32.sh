#!/bin/bash
# round 6, call 32: rotation of the exchange rows, 2 / 3 / 6 lines per plane on the other BASELINE shapes (four plans per point)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
L=$OUT/rot_lines_other_shapes.log
: > $L
for rl in 3 2 6 3 2 6; do
  echo "## DFFT_ROT_LINES=$rl" >> $L
  DFFT_ROT_LINES=$rl python tools/local_by_P.py 512x512x512 fp64 4 2,4,8 2>&1 | grep "rot=1" | sed 's/^/512^3 fp64   /' >> $L
  DFFT_ROT_LINES=$rl python tools/local_by_P.py 1024x768x512 fp64 4 4,8 2>&1 | grep "rot=1" | sed 's/^/config 4     /' >> $L
  DFFT_ROT_LINES=$rl python tools/local_by_P.py 1024x1024x1024 fp64 3 8 serial 2>&1 | grep "rot=1" | sed 's/^/1024^3 fp64  /' >> $L
  DFFT_ROT_LINES=$rl python tools/local_by_P.py 512x512x512 fp32 4 4 serial 2>&1 | grep "rot=1" | sed 's/^/512^3 fp32   /' >> $L
done
cat $L
