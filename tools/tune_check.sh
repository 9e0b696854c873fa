#!/bin/bash
# Does the plan-time placement measurement land bench.py in the fast X-pass mode?  N bench processes with the tuning's own
# report (bench JSON "plan_tune"), then the drop-in CLI on the same size (its "Forward FFT time" is one host-timed transform).
OUT=${1:-gpurun_out/r03/tune_check.log}
N=${2:-6}
mkdir -p "$(dirname "$OUT")"
for i in $(seq 1 $N); do
    python bench.py --no-cpu-baseline --no-pmc --steps 20 --warmup 5 2>/tmp/tune_check.err |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('run $i  ms/step %.4f  t0 %.4f  t3 %.4f  plan_tune %s' % (d['ms_per_step'], d['stages_ms']['t0'], d['stages_ms']['t3'], json.dumps(d.get('plan_tune'))))" | tee -a "$OUT"
done
for i in 1 2 3; do
    sh speedTest.sh 1 512 512 512 2>&1 | grep -E "^t0:|Forward FFT time|Performance" | tr '\n' ' ' | tee -a "$OUT"; echo | tee -a "$OUT"
done
