#!/bin/bash
# round 6, call 38: cross-lane (gfx950 v_permlane32/16_swap + DPP) against LDS for the wave-local 8 x 8 exchange of 16-byte elements
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
timeout 300 tools/bin/xlane_probe 4000 > $OUT/xlane_probe.log 2>&1; echo "exit $?" >> $OUT/xlane_probe.log
cat $OUT/xlane_probe.log
timeout 900 python tools/fuzz_parity.py 32 11 28 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" > $OUT/fuzz_parity_seed11.log
tail -3 $OUT/fuzz_parity_seed11.log
