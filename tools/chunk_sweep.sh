#!/bin/bash
# Z+Y cache-chunk size sweep on the graded bench (512^3 fp64, P = 1): planes per chunk vs whole grid-stride rounds.
#   tools/chunk_sweep.sh [outfile]
# DFFT_CHUNK_ROUNDS=0 = even split (round 2 before this sweep: 9 x 57 planes), default = whole-round rule of dfft_plan_create,
# DFFT_CHUNK_PLANES=n = explicit size (remainder chunk first; DFFT_CHUNK_SMALL_LAST=1: last).
OUT=${1:-gpurun_out/r02/chunk_sweep.log}
mkdir -p "$(dirname "$OUT")"
run() {
    local label="$1"; shift
    env "$@" python bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/tmp/chunk_sweep.err |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s ms/step %.4f  t0 %.4f  t3 %.4f' % ('$label', d['ms_per_step'], d['stages_ms']['t0'], d['stages_ms']['t3']))" | tee -a "$OUT"
    grep "Z+Y chunks" /tmp/chunk_sweep.err | head -1 | tee -a "$OUT"
}
echo "# $(date -u) chunk sweep, bench.py --steps 40 --warmup 5" | tee -a "$OUT"
run "even split (57)" DFFT_CHUNK_ROUNDS=0
run "whole-round rule" DFFT_DEBUG=1
run "56 first-small" DFFT_CHUNK_PLANES=56
run "56 last-small" DFFT_CHUNK_PLANES=56 DFFT_CHUNK_SMALL_LAST=1
run "60 first-small" DFFT_CHUNK_PLANES=60
run "52 first-small" DFFT_CHUNK_PLANES=52
run "48 first-small" DFFT_CHUNK_PLANES=48
run "64" DFFT_CHUNK_PLANES=64
run "40 first-small" DFFT_CHUNK_PLANES=40
run "32" DFFT_CHUNK_PLANES=32
run "even split (57) again" DFFT_CHUNK_ROUNDS=0
run "whole-round rule again" DFFT_DEBUG=1
