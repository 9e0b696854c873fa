#!/bin/bash
# round 6, call 25: parity of the library with the 256 MiB phase rule of packed forward launches; the same question for BACKWARD packed launches
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd $R
L=$OUT/packed_phase_backward.log
: > $L
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_multiprocess.py -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3 >> $L
for rep in 1 2; do
for cp in default 64; do
  echo "## backward, DFFT_CHUNK_PLANES=$cp" >> $L
  if [ $cp = default ]; then DFFT_AB_DIR=-1 python tools/lib_ab.py 512x512x512:fp64:4 512x512x512:fp64:2 2>&1 | grep -v amdgpu.ids | cut -c1-200 >> $L
  else DFFT_AB_DIR=-1 DFFT_CHUNK_PLANES=$cp python tools/lib_ab.py 512x512x512:fp64:4 512x512x512:fp64:2 2>&1 | grep -v amdgpu.ids | cut -c1-200 >> $L; fi
done
for cp in default 40; do
  echo "## backward, DFFT_CHUNK_PLANES=$cp" >> $L
  if [ $cp = default ]; then DFFT_AB_DIR=-1 python tools/lib_ab.py 1024x768x512:fp64:2 2>&1 | grep -v amdgpu.ids | cut -c1-200 >> $L
  else DFFT_AB_DIR=-1 DFFT_CHUNK_PLANES=$cp python tools/lib_ab.py 1024x768x512:fp64:2 2>&1 | grep -v amdgpu.ids | cut -c1-200 >> $L; fi
done
done
cat $L
