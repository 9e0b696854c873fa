#!/bin/bash
# round 6, call 37: radix-3 split tiles for 768-point columns (fft_dif3_tiles_kernel; -DDFFT_DIF3=1 build, DFFT_DIF3=1): parity, then config 4
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
export DFFT_LIB=$R/distributedfft_amd/lib/libdfft_variant_dif3.so
L=$OUT/lib_ab_768_dif3.log
: > $L
DFFT_DIF3=1 DFFT_T0_ONE_LAUNCH=0 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "768 or slab or SHAPES or rotated or exchange" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 >> $L
DFFT_DIF3=1 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "C4" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -2 >> $L
for rep in 1 2; do for d in 0 1; do
  echo "## DFFT_DIF3=$d" >> $L
  DFFT_DIF3=$d python tools/local_by_P.py 1024x768x512 fp64 3 8 2>&1 | grep "rot=1" >> $L
  DFFT_DIF3=$d DFFT_T0_ONE_LAUNCH=0 python tools/local_by_P.py 1024x768x512 fp64 3 1,4 serial 2>&1 | grep "rot=1\|P=1" | sed 's/^/two-launch  /' >> $L
done; done
cat $L
