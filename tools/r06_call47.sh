#!/bin/bash
# round 6, call 47: records of the library with the rows-first chunk loop of the inverse YZ stage (host-side plan change; device code unchanged):
# GPU suite, round trips, rocprofv3 stats + PMC passes of the graded bench command, the bench line, forward / backward status
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
sha256sum distributedfft_amd/lib/libdfft_mi355x_pt.so > $OUT/final_library.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" > $OUT/pytest_gpu_full.log; tail -3 $OUT/pytest_gpu_full.log
timeout 900 python tools/roundtrip_check.py 2>&1 | grep "^ok\|^FAIL\|shapes" > $OUT/roundtrip_check.log; cat $OUT/roundtrip_check.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
PROFILE_SKIP_NOCHUNK=1 timeout 1200 bash tools/profile_bench.sh r06 > $OUT/profile_bench.log 2>&1
cd $R
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err; cut -c1-300 $OUT/bench_driver_args.json
L=$OUT/backward_status_final.log; : > $L
SPECS="512x512x512:fp64:1 512x512x512:fp64:4 1024x768x512:fp64:8 1024x768x512:fp64:1 2048x2048x1024:fp32:8 1024x1024x1024:fp32:1 2048x1024x512:fp64:1 512x512x512:fp32:1 1024x768x512:fp32:1 1024x1024x1024:fp64:1"
for rep in 1 2; do
  echo "## forward" >> $L
  timeout 600 python tools/lib_ab.py $SPECS 2>&1 | grep "sha" | cut -c1-330 >> $L
  echo "## backward" >> $L
  DFFT_AB_DIR=-1 timeout 600 python tools/lib_ab.py $SPECS 2>&1 | grep "sha" | cut -c1-330 >> $L
done
grep -c sha $L; du -sh $R/gpurun_out
