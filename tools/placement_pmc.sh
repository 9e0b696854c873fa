#!/bin/bash
# Fast-vs-slow hand-over buffers of the 512^3 fp64 X pass: timings by allocation mode, then per-dispatch PMC counters.
# Each --pmc set is its own pass (with --kernel-trace only; gpurun refuses PMC + runtime traces).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/placement_r03; mkdir -p $OUT
NPL=${NPL:-6}; EX=${EX:-12}
python $R/tools/placement_pmc.py 4 16 malloc,vmm:2,vmm:64,vmm:0,vmm:0:2048,malloc > $OUT/modes.log 2>&1
tail -30 $OUT/modes.log
i=0
while read -r SET; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace -E $R/tools/extra_counters.yaml --pmc $SET --output-format csv -d $OUT/pass_$i -- python $R/tools/placement_pmc.py $NPL $EX > $OUT/pass_$i.log 2>&1
  echo "pass $i ($SET): rc $?"; grep "plan" $OUT/pass_$i.log | head -$NPL
done <<'SETS'
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum
DFFT_RDREQ_max DFFT_RDREQ_min DFFT_RDSTALL_max DFFT_RDSTALL_min
DFFT_WRREQ_max DFFT_WRREQ_min DFFT_TAGSTALL_max TCC_EA0_WRREQ_STALL_sum
TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE GRBM_EA_BUSY
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum
SETS
python $R/tools/pmc_summary.py $OUT $EX > $OUT/xpass_summary.txt 2>&1
python $R/tools/pmc_summary.py $OUT $((EX*8)) TuneStreamIn > $OUT/zpass_summary.txt 2>&1
python $R/tools/pmc_summary.py $OUT $((EX*8)) "TuneCols>" > $OUT/ypass_summary.txt 2>&1
find $OUT -name "*.csv" -size +8M -delete
find $OUT -name "*.db" -delete
du -sh $OUT
