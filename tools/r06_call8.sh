#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
L=$R/distributedfft_amd/lib
# config 4's rank at P = 8: the one-launch stage on 96-row destination blocks, now that its column units exchange differently
: > $OUT/c4_p8_one_launch_call8.log
for i in 1 2 3; do
  timeout 600 python tools/lib_ab.py 1024x768x512:fp64:8 1024x768x512:fp64:8:DFFT_T0_ONE_LAUNCH=all 1024x768x512:fp64:8:DFFT_T0_ONE_LAUNCH=all+DFFT_CHUNK_PLANES=43 2>&1 | grep -v amdgpu.ids >> $OUT/c4_p8_one_launch_call8.log
done
RECORDS_SKIP="pytest" bash tools/records.sh r06 > $OUT/records.log 2>&1
