#!/bin/bash
# round 6, call 24: interleaved ticket order of the one-launch YZ stage (DFFT_ZY_INTERLEAVE=1): rows of phase s alternate with columns of phase s - 1
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd $R
L=$OUT/zy_interleaved_order.log
: > $L
DFFT_ZY_INTERLEAVE=1 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "one_launch or fft2d" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3 >> $L
one() { # label size P env...
  lab=$1; sz=$2; P=$3; shift 3
  echo -n "$lab  " >> $L
  env "$@" python tools/local_by_P.py $sz fp64 2 $P serial 2>&1 | grep "rot=1\|P=1 " | head -1 >> $L
}
one "512^3 P=4 default        " 512x512x512 4 A=1
for cp in 16 20 24 28 32 43 64; do one "512^3 P=4 interleave cp=$cp" 512x512x512 4 DFFT_ZY_INTERLEAVE=1 DFFT_CHUNK_PLANES=$cp; done
one "512^3 P=1 default        " 512x512x512 1 A=1
for cp in 16 24 28 32 40 57; do one "512^3 P=1 interleave cp=$cp" 512x512x512 1 DFFT_ZY_INTERLEAVE=1 DFFT_CHUNK_PLANES=$cp; done
one "512^3 P=2 default        " 512x512x512 2 A=1
for cp in 26 32; do one "512^3 P=2 interleave cp=$cp" 512x512x512 2 DFFT_ZY_INTERLEAVE=1 DFFT_CHUNK_PLANES=$cp; done
one "c4 P=4 default           " 1024x768x512 4 A=1
for cp in 16 19 21; do one "c4 P=4 interleave cp=$cp" 1024x768x512 4 DFFT_ZY_INTERLEAVE=1 DFFT_CHUNK_PLANES=$cp; done
cat $L
