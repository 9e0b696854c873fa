#!/bin/bash
# round 6, call 34: plane pad of the single-GPU hand-over buffer in cache lines (default 3; round 4 swept odd values only): 1 2 3 6 10
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
L=$OUT/pad_plane_even.log
: > $L
for rep in 1 2; do for pp in 3 2 6 1 10; do
  echo "## DFFT_PAD_PLANE=$pp" >> $L
  DFFT_PAD_PLANE=$pp python tools/lib_ab.py 512x512x512:fp64:1 1024x768x512:fp64:1 2048x1024x512:fp32:1 1024x1024x1024:fp32:1 2048x1024x512:fp64:1 2>&1 | grep -v amdgpu.ids | cut -c30-175 >> $L
done; done
cat $L
