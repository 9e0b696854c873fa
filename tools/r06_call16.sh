#!/bin/bash
# round 6, call 16: config 4's overlapped t0 (+12-13 % over the serial plan at P = 4 / 8, profiles/r06/local_by_P.log) -- does the
# one-launch stage for all parts (32-row-granular destination blocks, PK2) pay there?  And the chunk size of the two-launch loop.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd $R
L=$OUT/c4_overlap_one_launch.log
: > $L
run() { echo "## $*" >> $L; env "$@" python tools/local_by_P.py 1024x768x512 fp64 3 4,8 2>&1 | grep -v amdgpu.ids >> $L; }
run A=default
run DFFT_T0_ONE_LAUNCH=all
run DFFT_T0_ONE_LAUNCH=all DFFT_OVERLAP_YPARTS=1
run DFFT_OVERLAP_YPARTS=1
for cp in 32 40 43 48 64; do
  echo "## DFFT_CHUNK_PLANES=$cp" >> $L
  DFFT_CHUNK_PLANES=$cp python tools/local_by_P.py 1024x768x512 fp64 3 8 2>&1 | grep -v amdgpu.ids | grep serial >> $L
done
