#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_call6.log 2>&1; tail -3 $OUT/pytest_gpu_call6.log
python tools/local_by_P.py 512x512x512 fp64 3 2>&1 | grep -v amdgpu.ids > $OUT/local_by_P_call6.log
DFFT_LIB=$R/distributedfft_amd/lib/libdfft_variant_sigsc1.so python tools/local_by_P.py 512x512x512 fp64 3 2>&1 | grep -v amdgpu.ids > $OUT/local_by_P_call6_sc1_only.log
for n in 2 4 8; do
  DFFT_BENCH_ALLOW_SHARED_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 5 --warmup 2 > $OUT/bench_shared_gpu_N$n.json 2> $OUT/bench_shared_gpu_N$n.err
done
