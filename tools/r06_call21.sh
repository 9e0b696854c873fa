#!/bin/bash
# round 6, call 21: row plans -- 1024 fp32 points on one wavefront (16 per thread), 1000 points on one wavefront (20 per thread)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd $R
V=$R/distributedfft_amd/lib/libdfft_variant_rows.so
L=$OUT/lib_ab_row_plans.log
: > $L
DFFT_LIB=$V python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "rows or 1000 or 1024 or fft1d" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3 >> $L
for rep in 1 2; do
  echo "## default library" >> $L
  python tools/long_axis_bench.py 2>/dev/null | grep "^rows\|^cols,1000\|1024x768x512,f32" >> $L
  python tools/lib_ab.py 2048x2048x1024:fp32:8 1000x1000x512:fp64:4 1024x1024x1024:fp32:1 2>&1 | grep -v amdgpu.ids | cut -c1-150 >> $L
  echo "## variant (DFFT_ROWS32_1024_E16=1 DFFT_ROWS_1000_E20=1)" >> $L
  DFFT_LIB=$V python tools/long_axis_bench.py 2>/dev/null | grep "^rows\|^cols,1000\|1024x768x512,f32" >> $L
  DFFT_LIB=$V python tools/lib_ab.py 2048x2048x1024:fp32:8 1000x1000x512:fp64:4 1024x1024x1024:fp32:1 2>&1 | grep -v amdgpu.ids | cut -c1-150 >> $L
done
cat $L
