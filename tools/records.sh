#!/bin/bash
# Everything the round's profiles/<round>/ directory records, in one GPU call (run through gpurun; raw output under gpurun_out/):
#   tools/records.sh [round, default r03]
# 1. rocprofv3 kernel trace + stats and FETCH_SIZE / WRITE_SIZE passes of the graded bench command (tools/profile_bench.sh)
# 2. two more counter passes (SQ wave-state counters; fabric request counts and queue levels) on the same command
# 3. 3D sweep, long-axis kernels, the reference's batched component benchmarks, per-rank local work by P, tuning check,
#    the bench line with the driver's arguments (CPU leg included), the full -m gpu suite
# RECORDS_SKIP="batch local pytest": leave those parts out (a refresh after a kernel change that does not touch them)
ROUND=${1:-r03}
skip() { case " $RECORDS_SKIP " in *" $1 "*) return 0;; esac; return 1; }
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/$ROUND; mkdir -p $OUT
cd $R
PROFILE_SKIP_NOCHUNK=1 bash tools/profile_bench.sh $ROUND > $OUT/profile_bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
P=$R/gpurun_out/prof_$ROUND
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
python tools/long_axis_bench.py > $OUT/long_axis_kernels.csv 2> /dev/null
bash tools/tune_check.sh $OUT/tune_check_final.log 6 > /dev/null 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
skip pytest || python -m pytest tests -m gpu -q > $OUT/pytest_gpu_full.log 2>&1
if ! skip local; then
python tools/local_by_P.py 512x512x512 fp64 4 2>&1 | grep -v amdgpu.ids > $OUT/local_by_P.log
python tools/local_by_P.py 1024x768x512 fp64 3 2>&1 | grep -v amdgpu.ids >> $OUT/local_by_P.log
python tools/local_by_P.py 2048x2048x1024 fp32 2 2>&1 | grep -v amdgpu.ids >> $OUT/local_by_P.log
fi
skip batch || NUM_ITER=100 bash tools/run_batch_tests.sh $OUT > /dev/null 2>&1
du -sh $OUT $P
