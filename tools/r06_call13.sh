#!/bin/bash
# round 6, session 2: staged store on the half-line tiles of the lean lengths -- parity of the new shapes, then the A/B of the X passes again
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06s2; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "slab_forward_backward or rotated_exchange" > $OUT/pytest_lean_shapes.log 2>&1; tail -3 $OUT/pytest_lean_shapes.log
SH="1536x1536x512:fp64:4 1536x1536x512:fp64:8 1536x1536x512:fp32:4 1000x1000x512:fp64:4 1000x1000x512:fp32:4 1280x1280x512:fp64:4 1000x1000x512:fp64:1 1000x1000x512:fp32:1 1280x1024x512:fp64:1 1280x1024x512:fp32:1 1536x1024x512:fp32:1"
for rep in 1 2; do
  for lib in default nolean; do
    if [ $lib = default ]; then unset DFFT_LIB; else export DFFT_LIB=$R/distributedfft_amd/lib/libdfft_variant_nolean.so; fi
    timeout 600 python tools/lib_ab.py $SH 2>&1 | sed "s/^/$lib  /" >> $OUT/lib_ab_lean_staged.log
  done
done
tail -2 $OUT/lib_ab_lean_staged.log
