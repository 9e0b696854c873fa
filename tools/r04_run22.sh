#!/bin/bash
# round-4 GPU call 22: the inverse YZ stage of 512^3 fp64 (1.37-1.41 ms against 1.155 forward): two launches per chunk, eager form, phase sizes
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp DFFT_AB_DIR=-1
S=512x512x512:fp64:1
( for rep in 1 2; do timeout 600 python tools/lib_ab.py $S $S:DFFT_T0_ONE_LAUNCH=0 $S:DFFT_ZY_LAZY=0 $S:DFFT_CHUNK_MB=200 $S:DFFT_CHUNK_MB=256 $S:DFFT_CHUNK_MB=170 $S:DFFT_CHUNK_MB=128 $S:DFFT_PAD=0 256x256x256:fp64:1 256x256x256:fp64:1:DFFT_T0_ONE_LAUNCH=0; done ) > $O/run22_backward_yz.log 2>&1
echo finished > $O/run22_done
