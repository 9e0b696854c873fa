#!/bin/bash
# round-4 GPU call 14: 1024-point transposing X pass on paired half-line tiles with TWO workgroups per CU (-DDFFT_DUAL_1024=1) against
# the shipped full-line tile with whole-tile prefetch: parity of every test that touches 1024 points, then A/B
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
L=distributedfft_amd/lib
( DFFT_LIB=$PWD/$L/libdfft_variant_dual1024.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | tail -6 ) > $O/run14_pytest_dual1024.log 2>&1
SPECS="1024x768x512:fp64:1 1024x768x512:fp32:1 1024x768x512:fp64:8 1024x768x512:fp64:4 1024x1024x1024:fp32:1 1024x1024x1024:fp64:8 1024x512x512:fp64:1 1024x256x256:fp64:1"
for rep in 1 2; do
  for lib in libdfft_mi355x_pt.so libdfft_variant_dual1024.so; do
    DFFT_LIB=$PWD/$L/$lib timeout 600 python tools/lib_ab.py $SPECS
  done
done > $O/run14_lib_ab_dual1024.log 2>&1
echo finished > $O/run14_done
