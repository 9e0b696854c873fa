#!/bin/bash
# round 6, call 36: bench.py after the referee change -- the default line (N = 1), and a shape whose overlapped plan selects another Y kernel (N = 8 sharing the GPU)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_after_referee.json 2> $OUT/bench_after_referee.err; echo rc=$?
python -c "
import json; d=json.loads(open('$OUT/bench_after_referee.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['cpu_baseline']['value'])"
DFFT_BENCH_ALLOW_SHARED_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29731 bench.py --gpus 8 --steps 3 --warmup 1 --size 384x1024x256 --precision fp32 > $OUT/bench_shared_gpu_referee_N8.json 2> $OUT/bench_shared_gpu_referee_N8.err; echo rc=$?
python -c "
import json; d=json.loads(open('$OUT/bench_shared_gpu_referee_N8.json').read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('ms_per_step','overlap_result_bit_identical','overlap_result_within_referee_tolerance','overlap_referee','overlap_fallback','pipeline','max_error')})"
