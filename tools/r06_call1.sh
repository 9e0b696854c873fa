#!/bin/bash
# round 6, GPU call 1: partition probe, full GPU suite on the thread-local-exchange build, A/B against the r05 library and the
# NOEXCH bound on the long-axis plans, counter passes for the limiter table
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
bash tools/partition_probe.sh > $OUT/partition_probe.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_call1.log 2>&1; tail -3 $OUT/pytest_gpu_call1.log
SPECS="1024x768x512:fp64:8 2048x2048x1024:fp32:8 1024x768x512:fp64:1 1024x768x512:fp32:8 1024x1024x1024:fp32:1 512x512x512:fp64:1"
L=$R/distributedfft_amd/lib
: > $OUT/lib_ab_local_exchange.log
for i in 1 2; do
  for lib in libdfft_variant_r05base.so libdfft_mi355x_pt.so libdfft_variant_noexch.so; do
    DFFT_LIB=$L/$lib timeout 600 python tools/lib_ab.py $SPECS 2>&1 | grep -v amdgpu.ids >> $OUT/lib_ab_local_exchange.log
  done
done
python tools/long_axis_bench.py > $OUT/long_axis_kernels_call1.csv 2> /dev/null
DFFT_LIB=$L/libdfft_variant_r05base.so python tools/long_axis_bench.py > $OUT/long_axis_kernels_r05base.csv 2> /dev/null
DFFT_LIB=$L/libdfft_variant_noexch.so python tools/long_axis_bench.py > $OUT/long_axis_kernels_noexch.csv 2> /dev/null
bash tools/long_axis_pmc.sh $OUT/long_axis_pmc > $OUT/long_axis_pmc.log 2>&1
