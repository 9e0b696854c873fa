#!/bin/bash
# round 6, session 2: experiment -- half-line tiles for the packed / rotated variants of 768 and 1024 points (-DDFFT_LEAN_EXTRA=1)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06s2; mkdir -p $OUT; cd $R
SH="1024x768x512:fp64:8 1024x768x512:fp64:4 1024x768x512:fp64:2 1024x768x512:fp32:8 768x768x768:fp64:4 1024x1024x1024:fp32:4 1024x1024x512:fp64:8"
for rep in 1 2 3; do
  for lib in default leanx; do
    if [ $lib = default ]; then unset DFFT_LIB; else export DFFT_LIB=$R/distributedfft_amd/lib/libdfft_variant_leanx.so; fi
    timeout 600 python tools/lib_ab.py $SH 2>&1 | sed "s/^/$lib  /" >> $OUT/lib_ab_lean_extra.log
  done
done
tail -2 $OUT/lib_ab_lean_extra.log
