#!/bin/bash
# round-4 GPU call 9: 768-point plan with 12 points x 64 threads (512-thread workgroups) against the shipped 24 x 32; chunk sizes whose
# column-tile count is a whole number of grid rounds (configs 4 / 5, two launches per chunk)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
L=distributedfft_amd/lib
( DFFT_LIB=$PWD/$L/libdfft_variant_e12.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "768 or slab or rotated or fft1d or rows or cols" 2>&1 | tail -5 ) > $O/run9_pytest_e12.log 2>&1
SPECS="1024x768x512:fp64:1 1024x768x512:fp32:1 1024x768x512:fp64:8 1024x768x512:fp64:4 768x768x768:fp64:1 768x768x768:fp32:1 512x768x512:fp64:1 512x512x768:fp64:1"
for rep in 1 2; do
  for lib in libdfft_mi355x_pt.so libdfft_variant_e12.so; do
    DFFT_LIB=$PWD/$L/$lib timeout 600 python tools/lib_ab.py $SPECS
  done
done > $O/run9_lib_ab_e12.log 2>&1
( timeout 600 python tools/variant_ab.py \
   "1024x768x512:fp64:1:2:c41=,c40=DFFT_CHUNK_PLANES=40,c36=DFFT_CHUNK_PLANES=36,c32=DFFT_CHUNK_PLANES=32,c42=DFFT_CHUNK_PLANES=42" \
   "2048x2048x1024:fp32:8:2:c15=,c12=DFFT_CHUNK_PLANES=12,c16=DFFT_CHUNK_PLANES=16,c14=DFFT_CHUNK_PLANES=14,c8=DFFT_CHUNK_PLANES=8" \
   "1024x768x512:fp64:8:2:c32=,c28=DFFT_CHUNK_PLANES=28,c40=DFFT_CHUNK_PLANES=40,c16=DFFT_CHUNK_PLANES=16" ) > $O/run9_chunk_rounds.log 2>&1
echo finished > $O/run9_done
