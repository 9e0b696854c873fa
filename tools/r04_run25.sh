#!/bin/bash
# round-4 GPU call 25: the multi-process file on the launcher that repeats a failed launch once
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
( timeout 200 python -m pytest tests/test_gpu_multiprocess.py -q -m gpu 2>&1 | tail -8 ) > $O/run25_pytest_multiprocess.log 2>&1
echo finished > $O/run25_done
