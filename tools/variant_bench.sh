run() { python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['stages_ms']['t0'], d['stages_ms']['t3'], d['direct_dft_spot_check_rel_error'])"; }
run base
for v in "$@"; do DFFT_LIB=$PWD/distributedfft_amd/lib/libdfft_variant_$v.so run $v; done
run base
