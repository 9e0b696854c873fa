#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_call4.log 2>&1; tail -3 $OUT/pytest_gpu_call4.log
L=$R/distributedfft_amd/lib
: > $OUT/lib_ab_backward_call4.log
for i in 1 2; do for lib in libdfft_variant_r05base.so libdfft_variant_nowo.so libdfft_mi355x_pt.so; do
  DFFT_AB_DIR=-1 DFFT_LIB=$L/$lib timeout 600 python tools/lib_ab.py 1024x768x512:fp64:8 2048x2048x1024:fp32:8 512x512x512:fp64:1 1024x768x512:fp64:1 512x512x512:fp64:4 256x256x256:fp64:1 1024x1024x1024:fp32:1 2>&1 | grep -v amdgpu.ids >> $OUT/lib_ab_backward_call4.log
done; done
: > $OUT/lib_ab_forward_call4.log
for i in 1 2; do for lib in libdfft_variant_r05base.so libdfft_mi355x_pt.so; do
  DFFT_LIB=$L/$lib timeout 600 python tools/lib_ab.py 1024x768x512:fp64:8 2048x2048x1024:fp32:8 512x512x512:fp64:1 768x768x768:fp64:1 1024x768x512:fp32:8 384x384x384:fp64:1 2>&1 | grep -v amdgpu.ids >> $OUT/lib_ab_forward_call4.log
done; done
NUM_ITER=100 bash tools/run_batch_tests.sh $OUT/batch_call4 > /dev/null 2>&1
