#!/bin/bash
# round 6, call 46: rows-first order for the inverse YZ stage's chunk loop (single-GPU plans with a hand-over buffer): -DDFFT_INV_CHUNK_ROWS_FIRST=1
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
V=$R/distributedfft_amd/lib/libdfft_variant_invrows.so
L=$OUT/lib_ab_inverse_rows_first_chunks.log
: > $L
SPECS="1024x1024x1024:fp32:1 2048x1024x512:fp64:1 1024x1024x1024:fp64:1 2048x1024x512:fp32:1 1024x768x512:fp32:1 512x512x512:fp32:1"
for rep in 1 2 3; do
  DFFT_AB_DIR=-1 timeout 600 python tools/lib_ab.py $SPECS 2>&1 | grep "sha" | cut -c1-200 >> $L
  DFFT_AB_DIR=-1 DFFT_LIB=$V timeout 600 python tools/lib_ab.py $SPECS 2>&1 | grep "sha" | cut -c1-200 >> $L
done
DFFT_LIB=$V timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3 >> $L
DFFT_LIB=$V timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3 >> $L
DFFT_LIB=$V timeout 900 python tools/roundtrip_check.py 2>&1 | grep "^ok\|^FAIL\|shapes" >> $L
cat $L
