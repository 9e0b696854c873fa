#!/bin/bash
# round 6, call 44: non-temporal stores of the staged transposed LOAD (inverse X pass of 2048-point column pairs):
# -DDFFT_TLOAD_NTS=1 build against the shipped library, backward plans
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
V=$R/distributedfft_amd/lib/libdfft_variant_tloadnts.so
L=$OUT/lib_ab_tload_nts.log
: > $L
SPECS="2048x2048x1024:fp32:8 2048x2048x1024:fp32:4 2048x1024x512:fp32:1 1024x1024x1024:fp32:1 1024x768x512:fp32:8 512x512x512:fp32:1 256x256x256:fp64:1"
for rep in 1 2 3; do
  DFFT_AB_DIR=-1 timeout 600 python tools/lib_ab.py $SPECS 2>&1 | grep "sha" | cut -c1-200 >> $L
  DFFT_AB_DIR=-1 DFFT_LIB=$V timeout 600 python tools/lib_ab.py $SPECS 2>&1 | grep "sha" | cut -c1-200 >> $L
done
DFFT_LIB=$V timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3 >> $L
DFFT_LIB=$V timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3 >> $L
cat $L
