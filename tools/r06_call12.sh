#!/bin/bash
# round 6, session 2: GPU suite on the spill-free build + A/B of the half-line ("lean") tiles against the full-line variants they replace
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06s2; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_full.log 2>&1; tail -3 $OUT/pytest_gpu_full.log
SH4="1536x1536x512:fp64:4 1536x1536x512:fp32:4 1280x1280x512:fp64:4 1280x1280x512:fp32:4 1000x1000x512:fp64:4 1000x1000x512:fp32:4 1536x1536x500:fp64:4 1000x1000x500:fp64:4 1536x1536x512:fp64:8"
SH1="1536x1024x512:fp64:1 1280x1024x512:fp64:1 1000x1000x512:fp64:1 1536x1024x512:fp32:1 1280x1024x512:fp32:1 1000x1000x512:fp32:1"
for rep in 1 2; do
  for lib in default nolean; do
    if [ $lib = default ]; then unset DFFT_LIB; else export DFFT_LIB=$R/distributedfft_amd/lib/libdfft_variant_nolean.so; fi
    timeout 600 python tools/lib_ab.py $SH4 2>&1 | sed "s/^/$lib  /" >> $OUT/lib_ab_lean.log
    timeout 600 python tools/lib_ab.py $SH1 2>&1 | sed "s/^/$lib  /" >> $OUT/lib_ab_lean.log
    DFFT_AB_DIR=-1 timeout 600 python tools/lib_ab.py 1536x1536x512:fp64:4 1280x1280x512:fp64:4 1000x1000x512:fp64:4 1280x1024x512:fp64:1 2>&1 | sed "s/^/$lib backward  /" >> $OUT/lib_ab_lean.log
  done
done
unset DFFT_LIB
# the scalar float2 fall-back: an odd fp32 Z length, and even shapes forced off the column pairs
timeout 600 python tools/lib_ab.py 512x512x511:fp32:1 512x512x512:fp32:1:DFFT_NO_PAIRS=1 > $OUT/scalar32.log 2>&1
timeout 600 python tools/lib_ab.py 512x512x512:fp32:1 1024x768x511:fp32:1 >> $OUT/scalar32.log 2>&1
DFFT_NO_PAIRS=1 timeout 600 python tools/lib_ab.py 1024x768x512:fp32:1 512x512x512:fp32:4 >> $OUT/scalar32.log 2>&1
tail -4 $OUT/lib_ab_lean.log
