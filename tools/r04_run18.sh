#!/bin/bash
# round-4 GPU call 18: staged transposed load for the inverse X pass (fft_tload_tiles_kernel): parity (every test with a backward plan),
# then backward plans against the build without it (-DDFFT_TLOAD=0)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
L=distributedfft_amd/lib
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | tail -25 ) > $O/run18_pytest.log 2>&1
export DFFT_AB_DIR=-1
S="512x512x512:fp64:1 512x512x512:fp32:1 256x256x256:fp64:1 1024x768x512:fp64:1 1024x768x512:fp32:1 1024x1024x1024:fp32:1 2048x1024x512:fp32:1 512x512x512:fp64:4 1024x768x512:fp64:8 2048x2048x1024:fp32:8 2048x2048x1024:fp32:4"
for rep in 1 2; do
  for lib in libdfft_mi355x_pt.so libdfft_variant_notload.so; do
    DFFT_LIB=$PWD/$L/$lib timeout 600 python tools/lib_ab.py $S
  done
done > $O/run18_lib_ab_tload.log 2>&1
echo finished > $O/run18_done
