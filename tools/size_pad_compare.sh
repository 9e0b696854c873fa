# bench.py on several sizes with and without the padded work buffer
run() { python bench.py --no-cpu-baseline --steps 50 --warmup 10 --size $2 --precision $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '$2', '$3', d['ms_per_step'], d['stages_ms']['t0'], d['stages_ms']['t3'])"; }
for sz in 256 384 512x256x256 256x512x512 512 1024x768x512; do for pr in fp64 fp32; do
  run default $sz $pr; DFFT_PAD=0 run nopad $sz $pr
done; done
