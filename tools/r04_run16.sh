#!/bin/bash
# round-4 GPU call 16: backward fp32 plans, fused (inverse X pass on the scalar fp32 column kernel: column pairs need adjacent columns on
# the input side) against the un-fused stage structure (rows + transpose) -- is the two-step X stage the faster one there?
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp DFFT_AB_DIR=-1
S="1024x1024x1024:fp32:1 2048x2048x1024:fp32:8 512x512x512:fp32:1 1024x768x512:fp32:1 2048x1024x512:fp32:1 512x512x512:fp64:1 1024x768x512:fp64:1"
( timeout 600 python tools/lib_ab.py $S; DFFT_AB_UNFUSED=1 timeout 600 python tools/lib_ab.py $(for s in $S; do echo $s:DFFT_AB_UNFUSED=1; done) ) > $O/run16_backward_fused_vs_unfused.log 2>&1
echo finished > $O/run16_done
