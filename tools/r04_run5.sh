#!/bin/bash
# round-4 GPU call 5: software-pipelined 2048-point tiles (paired half-line tiles of the X pass, DIF-split tiles of the Y pass):
# parity for every tile count, A/B against the un-pipelined build
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
L=distributedfft_amd/lib
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "2048 or pipelined or rotated_exchange or with_exchange_is_bit_identical or slab" 2>&1 | tail -30 ) > $O/run5_pytest.log 2>&1
SPECS="2048x1024x512:fp64:1 2048x1024x512:fp32:1 2048x2048x1024:fp32:8 512x2048x512:fp64:1 512x2048x512:fp32:1 2048x2048x256:fp32:1 2048x256x1024:fp64:1 2048x2048x1024:fp32:4"
for rep in 1 2; do
  for lib in libdfft_mi355x_pt.so libdfft_variant_nopipe.so; do
    DFFT_LIB=$PWD/$L/$lib timeout 600 python tools/lib_ab.py $SPECS
  done
done > $O/run5_lib_ab_pipeline.log 2>&1
echo finished > $O/run5_done
