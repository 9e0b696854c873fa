#!/bin/bash
# round-4 GPU call 3: the multi-process file first (call 2 hung in its 12-deep overlapped stress case for 600 s), then the rest of
# the suite, the read / write time-division copy probe, the bench line
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_multiprocess.py -q -m gpu 2>&1 | tail -60 ) > $O/run3_pytest_multiprocess.log 2>&1
( timeout 1200 python -m pytest tests -q -m gpu --ignore=tests/test_gpu_multiprocess.py 2>&1 | tail -40 ) > $O/run3_pytest_rest.log 2>&1
( timeout 300 tools/bin/tdm_copy 2 ) > $O/run3_tdm_copy.log 2>&1
( timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 ) > $O/run3_bench.json 2>&1
echo finished > $O/run3_done
