#!/bin/bash
# round 6, session 2: experiment -- the 1024-point forward X pass on DIF-split tiles with a staged store, two workgroups per CU (-DDFFT_X_DIF2_1024=1)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06s2; mkdir -p $OUT; cd $R
export DFFT_LIB=$R/distributedfft_amd/lib/libdfft_variant_xdif2.so
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "slab_forward_backward or rotated_exchange or 2048_point_tiles" > $OUT/pytest_xdif2.log 2>&1; tail -3 $OUT/pytest_xdif2.log
SH="1024x768x512:fp64:8 1024x768x512:fp64:4 1024x768x512:fp64:1 1024x768x512:fp32:8 1024x768x512:fp32:1 1024x1024x1024:fp32:1 1024x1024x1024:fp32:4 1024x1024x512:fp64:8 1024x512x512:fp64:2"
SH0=""; for s in $SH; do SH0="$SH0 $s:DFFT_X_DIF2=0"; done
for rep in 1 2 3; do
  timeout 600 python tools/lib_ab.py $SH 2>&1 | sed "s/^/dif2  /" >> $OUT/lib_ab_xdif2_1024.log
  DFFT_X_DIF2=0 timeout 600 python tools/lib_ab.py $SH 2>&1 | sed "s/^/ttf   /" >> $OUT/lib_ab_xdif2_1024.log
done
tail -2 $OUT/lib_ab_xdif2_1024.log
