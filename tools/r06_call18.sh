#!/bin/bash
# round 6, call 18: phase / chunk size of config 4's YZ stage per rank at P = 8 (128 planes of 6 MiB): the 32-plane chunk is slow on two
# launches (0.597 vs 0.549 ms for 40 planes) -- is it what makes the one-launch stage lose there (its 230 MiB rule gives 4 x 32)?
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd $R
L=$OUT/c4_p8_phase_sweep.log
: > $L
for cp in 22 26 28 30 32 34 36 37 38 40 43; do
  echo "## one launch, DFFT_CHUNK_PLANES=$cp" >> $L
  DFFT_T0_ONE_LAUNCH=all DFFT_CHUNK_PLANES=$cp python tools/local_by_P.py 1024x768x512 fp64 2 8 serial 2>&1 | grep "rot=1" >> $L
done
for cp in 26 30 34 36 38 40 42; do
  echo "## two launches, DFFT_CHUNK_PLANES=$cp" >> $L
  DFFT_CHUNK_PLANES=$cp python tools/local_by_P.py 1024x768x512 fp64 2 8 serial 2>&1 | grep "rot=1" >> $L
done
cat $L
