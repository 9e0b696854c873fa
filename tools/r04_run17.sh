#!/bin/bash
# round-4 GPU call 17: the chunk rule of fused P > 1 plans on BACKWARD plans (the A/B of call 10 was forward only)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp DFFT_AB_DIR=-1
S="1024x768x512:fp64:8 1024x768x512:fp64:4 2048x2048x1024:fp32:8 1024x1024x1024:fp64:8 2048x2048x1024:fp32:4"
for rep in 1 2; do
  timeout 600 python tools/lib_ab.py $S $(for s in $S; do echo $s:DFFT_CHUNK_RULE=0; done)
done > $O/run17_chunk_rule_backward.log 2>&1
echo finished > $O/run17_done
