#!/bin/bash
# round 6, call 22: phase size of PACKED one-launch YZ stages (P > 1: the Y side streams into the send buffer, the phase is only read from
# the cache) -- the 230 MiB head-room rule was measured on single-GPU in-place stages
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd $R
L=$OUT/packed_phase_sweep.log
: > $L
run() { # size P planes...
  sz=$1; P=$2; shift 2
  for cp in default "$@"; do
    echo "## $sz P=$P DFFT_CHUNK_PLANES=$cp" >> $L
    if [ $cp = default ]; then python tools/local_by_P.py $sz fp64 3 $P serial 2>&1 | grep "rot=1" >> $L
    else DFFT_CHUNK_PLANES=$cp python tools/local_by_P.py $sz fp64 3 $P serial 2>&1 | grep "rot=1" >> $L; fi
  done
}
run 1024x768x512 4 40 42 43 44 52 64
run 1024x768x512 2 40 43 52 64
run 512x512x512 2 57 60 64 66 86 128
run 512x512x512 4 32 64 66
cat $L
