#!/bin/bash
# round 6, call 33: library with the 2-line rotation rule: parity, then per-rank local work of the shapes it touches
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
L=$OUT/rot_rule_after.log
: > $L
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_multiprocess.py -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3 >> $L
DFFT_ROT_LINES=2 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "rotated or exchange or overlap" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -2 >> $L
python tools/local_by_P.py 2048x2048x1024 fp32 3 4,8 2>&1 | grep "rot=1\|^#" >> $L
python tools/local_by_P.py 1024x768x512 fp64 3 2,4,8 2>&1 | grep "rot=1\|^#" >> $L
python tools/local_by_P.py 1024x1024x1024 fp64 3 4,8 2>&1 | grep "rot=1\|^#" >> $L
cat $L
