#!/bin/bash
# t0 against the planes per Infinity-Cache phase of the one-launch YZ stage (DFFT_CHUNK_PLANES), graded bench, one process per setting
for c in ${@:-64 60 57 52 48 44 40 62 58 54 50 60 64 48}; do DFFT_CHUNK_PLANES=$c python bench.py --no-cpu-baseline --no-pmc --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunk $c  ms/step %.4f t0 %.4f t3 %.4f' % (d['ms_per_step'], d['stages_ms']['t0'], d['stages_ms']['t3']))"; done
