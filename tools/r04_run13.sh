#!/bin/bash
# round-4 GPU call 13: BACKWARD plans with 1024-point column passes (config 4's inverse X pass among them) with and without the
# DIF-split kernel from 1024 points on
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp DFFT_AB_DIR=-1
S2="1024x768x512:fp64:1 1024x768x512:fp64:8 1024x768x512:fp32:1 1024x1024x1024:fp32:1 512x1024x512:fp64:1 1024x512x512:fp64:4"
for rep in 1 2; do
  timeout 600 python tools/lib_ab.py $S2
  DFFT_DIF2_MIN=1024 timeout 600 python tools/lib_ab.py $(for s in $S2; do echo $s:DFFT_DIF2_MIN=1024; done)
done > $O/run13_dif2_1024_backward.log 2>&1
echo finished > $O/run13_done
