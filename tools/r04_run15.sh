#!/bin/bash
# round-4 GPU call 15: records of the final library (DIF-split kernel from 1024 points on): full GPU suite, profile of the bench command,
# bench line, sweeps; backward plans of the BASELINE shapes for the record
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
RECORDS_SKIP="batch local" bash tools/records.sh r04 > $O/records_final.log 2>&1
( DFFT_AB_DIR=-1 timeout 600 python tools/lib_ab.py 512x512x512:fp64:1 256x256x256:fp64:1 512x512x512:fp64:4 1024x768x512:fp64:8 2048x2048x1024:fp32:8 1024x1024x1024:fp32:1 ) > $O/run15_backward_plans.log 2>&1
( timeout 600 python tools/lib_ab.py 512x512x512:fp64:1 256x256x256:fp64:1 512x512x512:fp64:4 1024x768x512:fp64:8 2048x2048x1024:fp32:8 1024x1024x1024:fp32:1 ) > $O/run15_forward_plans.log 2>&1
echo finished > $O/run15_done
