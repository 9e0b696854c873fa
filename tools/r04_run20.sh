#!/bin/bash
# round-4 GPU call 20: bench processes with the longer placement walk (bench.py raises the walk's budget for its own process)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
rm -f $O/tune_check_final.log; bash tools/tune_check.sh $O/tune_check_final.log 10 > /dev/null 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
echo finished > $O/run20_done
