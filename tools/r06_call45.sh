#!/bin/bash
# round 6, call 45: counters on the kernels of the BACKWARD plans of configs 4 / 5 (rank 0 of P = 8, exchange off): what the inverse X pass of
# 2048-point pairs (staged transposed load on half-line tiles) does to the memory system, next to the forward kernels of limiter_table.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/r06/backward_pmc; mkdir -p $OUT
export DFFT_AB_DIR=-1
PROBE="python $R/tools/lib_ab.py 2048x2048x1024:fp32:8 1024x768x512:fp64:8 1024x1024x1024:fp32:1"
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pass_$name -- $PROBE > $OUT/pass_$name.log 2>&1; }
run sq  SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU
run ea  TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum
run tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +20M -delete
python $R/tools/long_axis_pmc_table.py $OUT > $OUT/limiter_table.md 2> $OUT/limiter_table.err
cut -c1-260 $OUT/limiter_table.md; tail -3 $OUT/limiter_table.err; du -sh $OUT
