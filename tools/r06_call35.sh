#!/bin/bash
# round 6, call 35: does rotating the exchange rows pay where received planes are an odd multiple of 128 KiB apart (the rule asks for 256 KiB)?
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
L=$OUT/rot_pays_128k.log
: > $L
python tools/local_by_P.py 1024x768x512 fp64 3 8 2>&1 | grep -v amdgpu >> $L
python tools/local_by_P.py 1024x384x512 fp64 3 8 serial 2>&1 | grep -v amdgpu >> $L
python tools/local_by_P.py 1024x768x256 fp64 3 8 serial 2>&1 | grep -v amdgpu >> $L
python tools/local_by_P.py 512x512x512 fp32 3 8 2>&1 | grep -v amdgpu >> $L
python tools/local_by_P.py 2048x384x512 fp64 3 8 serial 2>&1 | grep -v amdgpu >> $L
python tools/local_by_P.py 512x768x512 fp64 3 8 2>&1 | grep -v amdgpu >> $L
python tools/local_by_P.py 1024x640x512 fp64 3 8 serial 2>&1 | grep -v amdgpu >> $L
cat $L
