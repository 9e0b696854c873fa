#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "2048 or fullsize or slab_forward or rotated or fft1d_cols" > $OUT/pytest_gpu_call9.log 2>&1; tail -3 $OUT/pytest_gpu_call9.log
: > $OUT/lib_ab_x_dif2.log
SPECS="2048x2048x1024:fp32:8 2048x1024x512:fp32:1 2048x512x512:fp32:1 2048x256x1024:fp32:4 2048x2048x1024:fp32:4"
for i in 1 2 3; do
  DFFT_X_DIF2=0 DFFT_LIB=$R/distributedfft_amd/lib/libdfft_mi355x_pt.so timeout 600 python tools/lib_ab.py $SPECS 2>&1 | grep -v amdgpu.ids | sed 's/^libdfft_mi355x_pt.so  /paired-half-line-tiles /' >> $OUT/lib_ab_x_dif2.log
  timeout 600 python tools/lib_ab.py $SPECS 2>&1 | grep -v amdgpu.ids >> $OUT/lib_ab_x_dif2.log
done
