#!/bin/bash
# round 6, call 23: phase-size curve of the packed one-launch stage, 512^3 fp64 per rank at P = 4 (128 planes of 4 MiB)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd $R
L=$OUT/packed_phase_curve_512_P4.log
: > $L
for cp in 8 16 20 24 28 30 31 32 33 34 36 40 43 44 48 52 56 60 62 63 64 65 66 72 128; do
  echo -n "planes=$cp  " >> $L
  DFFT_CHUNK_PLANES=$cp python tools/local_by_P.py 512x512x512 fp64 2 4 serial 2>&1 | grep "rot=1" >> $L
done
cat $L
