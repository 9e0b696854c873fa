#!/bin/bash
# Counter passes over tools/long_axis_probe.py (GPU box): wave-state, LDS, fabric latency, L2, L1 and HBM-byte
# counters per kernel of config 4's and config 5's rank at P = 8, with the 512^3 plan as the control.  Each pass is its own process
# (rocprofv3 --pmc with --kernel-trace only).   tools/long_axis_pmc.sh [outdir] [DFFT_LIB]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=${1:-$R/gpurun_out/r06/long_axis_pmc}; mkdir -p $OUT
[ -n "$2" ] && export DFFT_LIB=$2
PROBE="python $R/tools/long_axis_probe.py 4"
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pass_$name -- $PROBE > $OUT/pass_$name.log 2>&1; }
run sq  SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU
run ea  TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum
run tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +20M -delete
python $R/tools/long_axis_pmc_table.py $OUT > $OUT/limiter_table.md 2> $OUT/limiter_table.err
du -sh $OUT
