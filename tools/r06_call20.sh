#!/bin/bash
# round 6, call 20: packing Y pass of 2048-point column pairs on half-line tiles, two workgroups per CU (-DDFFT_DIF2_HALF=1 build, DFFT_Y_DIF2_HALF=1)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd $R
export DFFT_LIB=$R/distributedfft_amd/lib/libdfft_variant_half.so
L=$OUT/lib_ab_y_pass_2048_half_line.log
: > $L
DFFT_Y_DIF2_HALF=1 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "2048 or rotated or fullsize" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5 >> $L
for rep in 1 2 3; do
  for h in 0 1; do
    echo "## DFFT_Y_DIF2_HALF=$h" >> $L
    DFFT_Y_DIF2_HALF=$h python tools/lib_ab.py 2048x2048x1024:fp32:8 2048x2048x1024:fp32:4 2048x1024x512:fp32:2 2>&1 | grep -v amdgpu.ids >> $L
  done
done
cat $L
