#!/bin/bash
# round 6, call 27: bench.py --gpus N report shape with the final library (N ranks sharing the one GPU through hipIpc: a shape check,
# not a measurement), 512^3 and config 4
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
for n in 2 4 8; do
  DFFT_BENCH_ALLOW_SHARED_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2961$n bench.py --gpus $n --steps 5 --warmup 2 > $OUT/bench_shared_gpu_final_N$n.json 2> $OUT/bench_shared_gpu_final_N$n.err
  echo "N=$n rc=$?"; python - <<PY
import json
d=json.loads(open("$OUT/bench_shared_gpu_final_N$n.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ms_per_step","overlap_result_bit_identical","pipeline","exchange_fallback")}, d["config"].get("plan"), d["config"].get("exchange"))
PY
done
for n in 4 8; do
  DFFT_BENCH_ALLOW_SHARED_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2962$n bench.py --gpus $n --steps 3 --warmup 1 --size 1024x768x512 > $OUT/bench_shared_gpu_final_c4_N$n.json 2> $OUT/bench_shared_gpu_final_c4_N$n.err
  echo "c4 N=$n rc=$?"; python - <<PY
import json
d=json.loads(open("$OUT/bench_shared_gpu_final_c4_N$n.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ms_per_step","overlap_result_bit_identical","pipeline","exchange_fallback")}, d["config"].get("plan"), d["config"].get("exchange"))
PY
done
