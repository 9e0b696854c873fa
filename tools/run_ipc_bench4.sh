cd $GRAFT_REPO_ROOT
for r in 0 1 2 3; do
  RANK=$r LOCAL_RANK=$r WORLD_SIZE=4 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 DFFT_EXCHANGE=ipc python bench.py --gpus 4 --size 256 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_ipc4_$r.json 2> gpurun_out/bench_ipc4_$r.err &
done
wait
cat gpurun_out/bench_ipc4_0.json
