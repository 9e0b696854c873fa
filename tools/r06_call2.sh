#!/bin/bash
# round 6, GPU call 2: wave-owned exchange -- full GPU suite, then A/B against the r05 library and the build without the labelling
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_call3.log 2>&1; tail -3 $OUT/pytest_gpu_call2.log
SPECS="1024x768x512:fp64:8 2048x2048x1024:fp32:8 1024x768x512:fp64:1 512x512x512:fp64:4 512x512x512:fp64:1 256x256x256:fp64:1 1024x1024x1024:fp32:1 2048x1024x512:fp64:1"
L=$R/distributedfft_amd/lib
: > $OUT/lib_ab_wave_owned_swz.log
for i in 1 2 3; do
  for lib in libdfft_variant_r05base.so libdfft_variant_nowo.so libdfft_mi355x_pt.so; do
    DFFT_LIB=$L/$lib timeout 600 python tools/lib_ab.py $SPECS 2>&1 | grep -v amdgpu.ids >> $OUT/lib_ab_wave_owned_swz.log
  done
done
: > $OUT/lib_ab_wave_owned_swz_backward.log
for lib in libdfft_variant_r05base.so libdfft_mi355x_pt.so libdfft_variant_r05base.so libdfft_mi355x_pt.so; do
  DFFT_AB_DIR=-1 DFFT_LIB=$L/$lib timeout 600 python tools/lib_ab.py 1024x768x512:fp64:8 2048x2048x1024:fp32:8 512x512x512:fp64:1 1024x768x512:fp64:1 2>&1 | grep -v amdgpu.ids >> $OUT/lib_ab_wave_owned_swz_backward.log
done
for lib in libdfft_variant_r05base.so libdfft_mi355x_pt.so libdfft_variant_r05base.so libdfft_mi355x_pt.so; do
  echo "== $lib" >> $OUT/long_axis_kernels_call3.csv
  DFFT_LIB=$L/$lib python tools/long_axis_bench.py 2> /dev/null | head -26 >> $OUT/long_axis_kernels_call3.csv
done
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_call3.json 2> $OUT/bench_call3.err; cat $OUT/bench_call3.json | cut -c1-400
