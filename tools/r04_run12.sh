#!/bin/bash
# round-4 GPU call 12: the parity file with 1024-point column passes on the DIF-split kernel (DFFT_DIF2_MIN=1024) -- is it safe as the
# default?; the multi-process file three more times (how often does the 4-rank stress case stall?)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
( DFFT_DIF2_MIN=1024 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_batch_tools.py -q -m gpu 2>&1 | tail -8 ) > $O/run12_pytest_dif2_1024.log 2>&1
for i in 1 2 3; do ( timeout 900 python -m pytest tests/test_gpu_multiprocess.py -q -m gpu 2>&1 | tail -4 ); done > $O/run12_pytest_multiprocess_x3.log 2>&1
echo finished > $O/run12_done
