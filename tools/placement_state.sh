#!/bin/bash
# Does the pattern of fast / slow buffer pairs depend on what ran on the GPU before?  Census of consecutive 2 GiB allocations
# (tools/xprobe.hip, mode -1) on the fresh box, after a process that used 3 x 32 GiB buffers, and after a few minutes of the
# GPU test suite; the tuning check (fresh bench.py processes) at the end.     tools/placement_state.sh <log>
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
LOG=${1:-$R/gpurun_out/r03/placement_state.log}; mkdir -p $(dirname $LOG)
cd $R
N=${CENSUS:-80}
echo "### fresh box" > $LOG
tools/bin/xprobe -1 $N >> $LOG 2>&1
echo "### after local_by_P 2048x2048x1024 fp32 (3 x 32 GiB buffers, plans created and destroyed)" >> $LOG
python tools/local_by_P.py 2048x2048x1024 fp32 1 > /dev/null 2>&1
tools/bin/xprobe -1 $N >> $LOG 2>&1
echo "### after the parity tests of the tuned lengths and the 3D shapes (about a minute of plans of all sizes)" >> $LOG
timeout 240 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "tuned or 3d or rotated" > /dev/null 2>&1
tools/bin/xprobe -1 $N >> $LOG 2>&1
bash tools/tune_check.sh $LOG.tune 4 > /dev/null 2>&1
echo "### tuning check afterwards" >> $LOG
cat $LOG.tune >> $LOG; rm -f $LOG.tune
