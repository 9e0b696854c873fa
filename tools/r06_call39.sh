#!/bin/bash
# round 6, call 39: the wave-owned exchange across lanes (v_permlane32/16_swap + DPP; -DDFFT_XLANE=1 build) against the shipped LDS form
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
V=$R/distributedfft_amd/lib/libdfft_variant_xlane.so
L=$OUT/lib_ab_xlane.log
: > $L
SPECS="1024x768x512:fp64:8 2048x2048x1024:fp32:8 1024x768x512:fp64:1 512x512x512:fp64:4 512x512x512:fp64:1 1024x1024x1024:fp32:1 2048x1024x512:fp64:1 1024x1024x512:fp64:8"
for rep in 1 2 3; do
  timeout 600 python tools/lib_ab.py $SPECS 2>&1 | grep "sha" | cut -c1-250 >> $L
  DFFT_LIB=$V timeout 600 python tools/lib_ab.py $SPECS 2>&1 | grep "sha" | cut -c1-250 >> $L
done
DFFT_LIB=$V timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3 >> $L
cat $L
