#!/bin/bash
# Developer helper (GPU box): bench.py with 4 ranks sharing the one GPU through the hipIpc communicator.
#   EX=ipc|ipc-async SIZE=256 bash tools/run_ipc_bench4.sh
cd ${GRAFT_REPO_ROOT:-$(dirname "$(readlink -f "$0")")/..}
mkdir -p gpurun_out
EX=${EX:-ipc}; SIZE=${SIZE:-256}; W=${W:-4}
for ((r = 0; r < W; r++)); do
  RANK=$r LOCAL_RANK=$r WORLD_SIZE=$W MASTER_ADDR=127.0.0.1 MASTER_PORT=${PORT:-29611} DFFT_EXCHANGE=$EX \
    python bench.py --gpus $W --size $SIZE --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_${EX}_$r.json 2> gpurun_out/bench_${EX}_$r.err &
done
wait
cat gpurun_out/bench_${EX}_0.json
