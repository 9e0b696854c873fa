#!/bin/bash
# round-4 GPU call 7: whole-tile prefetch + early wait (DFFT_WIDE_PREFETCH=2) for the 768-point column kernels against the shipped build;
# the 2048-point tile-count test on the shipped build
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
L=distributedfft_amd/lib
( timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "tile_count or 768 or slab" 2>&1 | tail -5 ) > $O/run7_pytest.log 2>&1
( DFFT_LIB=$PWD/$L/libdfft_variant_wide2.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "768 or 384 or slab or rotated" 2>&1 | tail -5 ) > $O/run7_pytest_wide2.log 2>&1
SPECS="1024x768x512:fp64:1 1024x768x512:fp32:1 1024x768x512:fp64:8 1024x768x512:fp64:4 768x768x768:fp64:1 768x768x768:fp32:1 512x768x512:fp64:1 384x384x384:fp64:1"
for rep in 1 2; do
  for lib in libdfft_mi355x_pt.so libdfft_variant_wide2.so; do
    DFFT_LIB=$PWD/$L/$lib timeout 600 python tools/lib_ab.py $SPECS
  done
done > $O/run7_lib_ab_wide2.log 2>&1
echo finished > $O/run7_done
