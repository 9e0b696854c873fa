#!/bin/bash
# round-4 GPU call 6: whole-tile prefetch in AGPR-backed registers for the 256-thread big-tile column kernels (768 points: config 4's
# Y axis): parity of the whole parity file, A/B against the build without it
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
L=distributedfft_amd/lib
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -30 ) > $O/run6_pytest.log 2>&1
SPECS="1024x768x512:fp64:1 1024x768x512:fp32:1 1024x768x512:fp64:8 1024x768x512:fp64:4 768x768x768:fp64:1 768x768x768:fp32:1 512x768x512:fp64:1 384x384x384:fp64:1 512x384x512:fp64:1"
for rep in 1 2; do
  for lib in libdfft_mi355x_pt.so libdfft_variant_nowide.so; do
    DFFT_LIB=$PWD/$L/$lib timeout 600 python tools/lib_ab.py $SPECS
  done
done > $O/run6_lib_ab_wide_prefetch.log 2>&1
echo finished > $O/run6_done
