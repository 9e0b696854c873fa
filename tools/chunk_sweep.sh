run() { python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['stages_ms']['t0'], d['stages_ms']['t3'])"; }
run default
for mb in 240 224 200 261; do DFFT_CHUNK_MB=$mb run "chunk_mb=$mb"; done
DFFT_PAD_PLANE=5 run plane5
DFFT_PAD_PLANE=1 run plane1
run default
