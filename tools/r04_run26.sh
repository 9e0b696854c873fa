#!/bin/bash
# round-4 GPU call 26 (the last seconds of the budget): backward 512^3 fp64 on the row-pitch build of the one-launch stage, rows of the
# hand-over buffer padded by one line, against the shipped build; equal digests = the experiment build computes the same transform
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp DFFT_AB_DIR=-1
L=distributedfft_amd/lib
S=512x512x512:fp64:1
( timeout 40 python tools/lib_ab.py $S
  DFFT_LIB=$PWD/$L/libdfft_variant_rowpitch.so timeout 50 python tools/lib_ab.py $S $S:DFFT_PAD_ROW=1 $S:DFFT_PAD_ROW=1+DFFT_PAD_PLANE=0 ) > $O/run26_rowpitch.log 2>&1
echo finished > $O/run26_done
