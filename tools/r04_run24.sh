#!/bin/bash
# round-4 GPU call 24: the RCCL fall-back case that failed once in call 23, with its full report
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_gpu_multiprocess.py -q -m gpu -k "falls_back" 2>&1 | tail -120 ) > $O/run24_pytest_fallback.log 2>&1
echo finished > $O/run24_done
