#!/bin/bash
# round-4 GPU call 19: placement walk with the confirmed stop rule: 10 bench processes, the tune tests, then the profile of the final library
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "tune" 2>&1 | tail -3 ) > $O/run19_pytest_tune.log 2>&1
rm -f $O/tune_check_final.log; bash tools/tune_check.sh $O/tune_check_final.log 10 > /dev/null 2>&1
RECORDS_SKIP="batch local pytest" bash tools/records.sh r04 > $O/records_final.log 2>&1
echo finished > $O/run19_done
