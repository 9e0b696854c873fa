O=gpurun_out/r03b; mkdir -p $O
timeout 400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_full.log
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc"
for cp in 0 47 52 64; do
  if [ $cp = 0 ]; then timeout 120 $B > $O/bench_cp_default.json 2> $O/bench_cp_default.err
  else DFFT_CHUNK_PLANES=$cp timeout 120 $B > $O/bench_cp_$cp.json 2> $O/bench_cp_$cp.err; fi
done
DFFT_ZY_LAZY=0 timeout 120 $B > $O/bench_eager.json 2> $O/bench_eager.err
timeout 120 $B > $O/bench_cp_default2.json 2> $O/bench_cp_default2.err
tail -4 $O/pytest_gpu_full.log
for f in $O/bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["ms_per_step"], d["value"], d["stages_ms"], d["config"]["plan"], d.get("plan_tune"))
except Exception as e: print("ERR", e)
PY
done
