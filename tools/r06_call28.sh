#!/bin/bash
# round 6, call 28: batched in-place rows (Test_1D) with non-temporal loads (DFFT_ROWS_STREAM=1) against the default
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
BIN=$R/distributedfft_amd/lib
L=$OUT/rows_stream_ab.log
: > $L
for rep in 1 2; do for s in 0 1; do
  export DFFT_BATCH_CSV=$OUT/rows_stream_$s.csv; : > $DFFT_BATCH_CSV
  for X in 256 512 1024 2048 4096 243 625; do DFFT_ROWS_STREAM=$s $BIN/Test_1D $X 1 1 100 0 > /dev/null; done
  echo "## DFFT_ROWS_STREAM=$s" >> $L; cat $DFFT_BATCH_CSV >> $L
done; done
cat $L
