run() { python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['stages_ms']['t0'], d['stages_ms']['t3'])"; }
for mb in 256 261 300 330 224 192 128; do DFFT_CHUNK_MB=$mb run "chunk_mb=$mb"; done
DFFT_PAD=0 run nopad
