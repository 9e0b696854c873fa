run() { python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['stages_ms']['t0'], d['stages_ms']['t3'])"; }
for i in 1 2 3; do for cfg in "0 3" "0 5" "1 3"; do set -- $cfg; DFFT_PAD_ROW=$1 DFFT_PAD_PLANE=$2 run "row=$1 plane=$2"; done; done
DFFT_PAD=0 run nopad
