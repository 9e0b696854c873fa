#!/bin/bash
# round 6, call 41: rotation amount of the exchange rows for BACKWARD plans (the inverse X pass WRITES the rotated rows): DFFT_ROT_LINES swept
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
L=$OUT/rot_lines_backward.log
: > $L
S=""
for shape in 2048x2048x1024:fp32:8 1024x1024x1024:fp64:8 1024x768x512:fp64:8 512x512x512:fp64:4 1024x1024x1024:fp32:4; do
  S="$S $shape:DFFT_ROT=0+DFFT_ROT_LINES=0"
  for n in 1 2 3 4 6; do S="$S $shape:DFFT_ROT=1+DFFT_ROT_LINES=$n"; done
done
for rep in 1 2; do
  DFFT_AB_DIR=-1 timeout 900 python tools/lib_ab.py $S 2>&1 | grep "sha" | cut -c1-200 >> $L
done
cat $L
