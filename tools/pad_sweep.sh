run() { python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['stages_ms']['t0'], d['stages_ms']['t3'])"; }
for pl in 3 5 7 9 11 13 3 5; do DFFT_CHUNK_MB=261 DFFT_PAD_PLANE=$pl run "8x64 plane=$pl"; done
for pl in 3 5 7; do DFFT_PAD_PLANE=$pl run "9x57 plane=$pl"; done
