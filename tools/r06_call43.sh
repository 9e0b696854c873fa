#!/bin/bash
# round 6, call 43: the GPU suite and the default bench line at the last commit of the round (library 295af95c, unchanged)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
sha256sum distributedfft_amd/lib/libdfft_mi355x_pt.so > $OUT/final_check.log
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 >> $OUT/final_check.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> $OUT/final_check.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/final_bench_line.json
cat $OUT/final_bench_line.json >> $OUT/final_check.log
cat $OUT/final_check.log | cut -c1-600
