#!/bin/bash
# round 6, call 29: phase size of dfft_fft2d_batch (in place on the caller's un-padded planes): 230 MiB rule against whole 256 MiB phases
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
BIN=$R/distributedfft_amd/lib
L=$OUT/fft2d_phase_planes.log
: > $L
export DFFT_BATCH_CSV=$OUT/fft2d_phase.csv
for rep in 1 2; do
for pp in default 43 52 57 60 63 64 65 86 128; do
  : > $DFFT_BATCH_CSV
  if [ $pp = default ]; then $BIN/Test_2D 512 512 1 100 0 > /dev/null; else DFFT_2D_PHASE_PLANES=$pp $BIN/Test_2D 512 512 1 100 0 > /dev/null; fi
  echo "512x512x256 planes/phase=$pp  $(cat $DFFT_BATCH_CSV)" >> $L
done
for pp in default 208 230 256 260; do
  : > $DFFT_BATCH_CSV
  if [ $pp = default ]; then $BIN/Test_2D 256 256 1 100 0 > /dev/null; else DFFT_2D_PHASE_PLANES=$pp $BIN/Test_2D 256 256 1 100 0 > /dev/null; fi
  echo "256x256x1024 planes/phase=$pp  $(cat $DFFT_BATCH_CSV)" >> $L
done
done
cat $L
