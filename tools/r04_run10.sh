#!/bin/bash
# round-4 GPU call 10: chunk rule of the two-launch t0 (largest chunk of whole grid rounds that fits the cache as it lies in the hand-over
# buffer) against the rule of rounds 2-3, over the shapes the old rule was tuned on and the BASELINE configs' per-rank shapes
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
V="old=DFFT_CHUNK_RULE=0,new="
( timeout 1500 python tools/variant_ab.py \
   "512x512x512:fp64:1:2:old=DFFT_CHUNK_RULE=0+DFFT_T0_ONE_LAUNCH=0,new=DFFT_T0_ONE_LAUNCH=0" \
   "512x512x512:fp32:1:2:$V" "384x384x384:fp64:1:2:$V" "768x768x768:fp64:1:2:$V" "1024x1024x1024:fp32:1:2:$V" "1024x1024x1024:fp64:1:2:$V" \
   "1024x768x512:fp64:1:2:$V" "1024x768x512:fp32:1:2:$V" "2048x1024x512:fp64:1:2:$V" "2048x1024x512:fp32:1:2:$V" "512x2048x512:fp64:1:2:$V" \
   "1024x768x512:fp64:8:2:$V" "1024x768x512:fp64:4:2:$V" "2048x2048x1024:fp32:8:2:$V" "2048x2048x1024:fp32:4:2:$V" \
   "512x512x512:fp64:2:2:old=DFFT_CHUNK_RULE=0+DFFT_T0_ONE_LAUNCH=0,new=DFFT_T0_ONE_LAUNCH=0" "1024x1024x1024:fp64:8:2:$V" "640x640x640:fp64:1:2:$V" ) > $O/run10_chunk_rule.log 2>&1
echo finished > $O/run10_done
