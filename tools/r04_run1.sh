#!/bin/bash
# round-4 GPU call 1: ADVICE fixes on the GPU, chunk sweep of the lazy one-launch t0, wide column tiles for 256-point Y axes,
# the two build-time experiments left by round 3 (stage-major twiddle table, lazy-publish packed stage)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
L=distributedfft_amd/lib
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_launch or plan_tune or back_to_back" 2>&1 | tail -15 ) > $O/run1_pytest.log 2>&1
( timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 ) > $O/run1_bench.json 2>&1
( timeout 600 python tools/variant_ab.py \
   "512x512x512:fp64:1:2:c57=,c52=DFFT_CHUNK_PLANES=52,c64=DFFT_CHUNK_PLANES=64,c47=DFFT_CHUNK_PLANES=47,c43=DFFT_CHUNK_PLANES=43" \
   "256x256x256:fp64:1:3:two=DFFT_T0_ONE_LAUNCH=0,one=DFFT_T0_ONE_LAUNCH=1" \
   "512x256x256:fp64:1:3:two=DFFT_T0_ONE_LAUNCH=0,one=DFFT_T0_ONE_LAUNCH=1" \
   "256x256x512:fp64:1:3:two=DFFT_T0_ONE_LAUNCH=0,one=DFFT_T0_ONE_LAUNCH=1" \
   "256x512x256:fp64:1:3:two=DFFT_T0_ONE_LAUNCH=0,one=DFFT_T0_ONE_LAUNCH=1" ) > $O/run1_variant_ab.log 2>&1
SPECS_TW="1024x768x512:fp64:1 1024x768x512:fp32:1 1024x768x512:fp64:8 2048x1024x512:fp64:1 2048x1024x512:fp32:1 2048x2048x1024:fp32:8 1024x1024x1024:fp32:1 512x512x512:fp64:1"
for rep in 1 2; do
  for lib in libdfft_mi355x_pt.so libdfft_variant_twsm.so; do
    DFFT_LIB=$PWD/$L/$lib timeout 600 python tools/lib_ab.py $SPECS_TW
  done
done > $O/run1_lib_ab_twsm.log 2>&1
SPECS_PK="512x512x512:fp64:2:DFFT_T0_ONE_LAUNCH=0 512x512x512:fp64:2:DFFT_T0_ONE_LAUNCH=all 512x512x512:fp64:4:DFFT_T0_ONE_LAUNCH=0 512x512x512:fp64:4:DFFT_T0_ONE_LAUNCH=all 512x512x512:fp64:8:DFFT_T0_ONE_LAUNCH=0 512x512x512:fp64:8:DFFT_T0_ONE_LAUNCH=all"
for rep in 1 2; do
  for lib in libdfft_mi355x_pt.so libdfft_variant_lazypk.so; do
    DFFT_LIB=$PWD/$L/$lib timeout 600 python tools/lib_ab.py $SPECS_PK
  done
done > $O/run1_lib_ab_lazypk.log 2>&1
echo finished > $O/run1_done
