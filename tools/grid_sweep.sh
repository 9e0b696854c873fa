run() { python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['stages_ms']['t0'], d['stages_ms']['t3'])"; }
run base
for g in 128 192 256 384; do DFFT_Y_GRID=$g run "Y$g"; done
for g in 128 192 384 512; do DFFT_X_GRID=$g run "X$g"; done
for g in 256 384 512 768; do DFFT_Z_GRID=$g run "Z$g"; done
run base
