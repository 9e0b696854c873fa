#!/bin/bash
# Repeats of the three candidate Z+Y chunk sizes on the graded bench, interleaved, with the placement tuning's own log line.
OUT=${1:-gpurun_out/r02/chunk_repeat.log}
mkdir -p "$(dirname "$OUT")"
run() {
    local label="$1"; shift
    env DFFT_DEBUG=1 "$@" python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/tmp/chunk_rep.err |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-22s ms/step %.4f  t0 %.4f  t3 %.4f' % ('$label', d['ms_per_step'], d['stages_ms']['t0'], d['stages_ms']['t3']), end='  ')" | tee -a "$OUT"
    grep "hand-over buffer placement" /tmp/chunk_rep.err | tail -1 | sed 's/.*placement: //' | tee -a "$OUT"
}
echo "# $(date -u) chunk repeat, bench.py --steps 20 --warmup 5" | tee -a "$OUT"
for i in 1 2 3 4; do
    run "64 planes" DFFT_CHUNK_PLANES=64
    run "even split (57)" DFFT_CHUNK_ROUNDS=0
    run "60 (first 32)" DFFT_CHUNK_PLANES=60
    run "63 (first 8)" DFFT_CHUNK_PLANES=63
done
