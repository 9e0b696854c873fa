#!/bin/bash
# The reference's runTest1D_opt.sh / runTest2D_opt.sh sweeps (templateFFT/batchTest) over Test_1D / Test_2D, restricted to the
# single-pass range of the library (lengths <= 4096); writes batch_result1D.csv / batch_result2D.csv into $1 (default gpurun_out).
OUT=${1:-gpurun_out}; mkdir -p $OUT
BIN=$(dirname "$(readlink -f "$0")")/../distributedfft_amd/lib
num_iter=${NUM_ITER:-200}
HDR='X,Y,Z,Buffer,hip_time,GFlops,num_iter,bandwidth,max error'
echo "$HDR" > $OUT/batch_result1D.csv
export DFFT_BATCH_CSV=$OUT/batch_result1D.csv
for ((X=256; X<=4096; X=X*2)); do $BIN/Test_1D $X 1 1 $num_iter 0 > /dev/null; done
for ((X=3; X<=2187; X=X*3)); do $BIN/Test_1D $X 1 1 $num_iter 0 > /dev/null; done
for ((X=5; X<=3125; X=X*5)); do $BIN/Test_1D $X 1 1 $num_iter 0 > /dev/null; done
for ((X=7; X<=2401; X=X*7)); do $BIN/Test_1D $X 1 1 $num_iter 0 > /dev/null; done
echo "$HDR" > $OUT/batch_result2D.csv
export DFFT_BATCH_CSV=$OUT/batch_result2D.csv
for ((X=2048; X>=128; X=X/2)); do for ((Y=2048; Y>=128; Y=Y/2)); do $BIN/Test_2D $X $Y 1 $num_iter 0 > /dev/null; done; done
for XY in "360 360" "243 243" "729 243"; do $BIN/Test_2D $XY 1 $num_iter 0 > /dev/null; done
cat $OUT/batch_result1D.csv $OUT/batch_result2D.csv
