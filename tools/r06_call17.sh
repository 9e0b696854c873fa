#!/bin/bash
# round 6, call 17: overlapped plans of config 4 -- one launch for all parts on 96-row blocks (P = 4), parts sized by the chunk rule (P = 8)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_multiprocess.py -m gpu -q -x > $OUT/pytest_call17.log 2>&1
tail -5 $OUT/pytest_call17.log
L=$OUT/c4_overlap_after.log
: > $L
python tools/local_by_P.py 1024x768x512 fp64 3 2,4,8 2>&1 | grep -v amdgpu.ids >> $L
python tools/local_by_P.py 1024x1024x512 fp64 2 4,8 2>&1 | grep -v amdgpu.ids >> $L
python tools/local_by_P.py 512x512x512 fp64 2 4,8 2>&1 | grep -v amdgpu.ids >> $L
cat $L
