#!/bin/bash
# round 6, call 19: finer phase sweep of the one-launch stage on config 4's rank at P = 8 (4 plans per point), two launches beside it
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd $R
L=$OUT/c4_p8_phase_sweep_fine.log
: > $L
for rep in 1 2; do
for cp in 40 42 43 44 46 48 52 64; do
  echo "## one launch, DFFT_CHUNK_PLANES=$cp" >> $L
  DFFT_T0_ONE_LAUNCH=all DFFT_CHUNK_PLANES=$cp python tools/local_by_P.py 1024x768x512 fp64 4 8 serial 2>&1 | grep "rot=" >> $L
done
echo "## two launches, default" >> $L
python tools/local_by_P.py 1024x768x512 fp64 4 8 serial 2>&1 | grep "rot=" >> $L
done
cat $L
