#!/bin/bash
# round-4 GPU call 21: profile of the bench command on the final library AND the final bench.py (placement walk with the process's own budget)
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r04; mkdir -p $OUT
cd $R
rm -rf $R/gpurun_out/prof_r04
PROFILE_SKIP_NOCHUNK=1 bash tools/profile_bench.sh r04 > $OUT/profile_bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
P=$R/gpurun_out/prof_r04
BENCH2="python $R/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $P/pass_sq -- $BENCH2 > $P/pass_sq.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum --output-format csv -d $P/pass_ea -- $BENCH2 > $P/pass_ea.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum --output-format csv -d $P/pass_tcp -- $BENCH2 > $P/pass_tcp.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum --output-format csv -d $P/pass_tcc -- $BENCH2 > $P/pass_tcc.log 2>&1
python $R/tools/pmc_summary.py $P 1000000 zy_chunk_kernel > $OUT/pmc_zy_kernel.txt 2>&1
python $R/tools/pmc_summary.py $P 1000000 TuneTransposedStore > $OUT/pmc_x_kernel.txt 2>&1
find $P -name "*.db" -delete; find $P -name "*kernel_trace.csv" -size +20M -delete
cd $R
python tools/sweep_bench.py 3d > $OUT/sweep_3d.csv 2> /dev/null
echo finished > $OUT/run21_done
