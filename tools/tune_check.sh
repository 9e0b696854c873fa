#!/bin/bash
# Does the plan-time placement measurement land bench.py in the fast X-pass mode?  N bench processes with the tuning's own log line.
OUT=${1:-gpurun_out/r02/tune_check.log}
mkdir -p "$(dirname "$OUT")"
for i in 1 2 3 4 5 6; do
    DFFT_DEBUG=1 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/tmp/tune_check.err |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('run $i  ms/step %.4f  t0 %.4f  t3 %.4f' % (d['ms_per_step'], d['stages_ms']['t0'], d['stages_ms']['t3']), end='  ')" | tee -a "$OUT"
    grep "hand-over buffer placement" /tmp/tune_check.err | tail -1 | sed 's/.*placement: //' | tee -a "$OUT"
done
