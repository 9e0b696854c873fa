#!/bin/bash
# round-4 GPU call 2: full GPU suite + the graded bench line on the tree with stage-major twiddles, lazy packed one-launch t0,
# wide column tiles for 256-point Y axes
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 ) > $O/run2_pytest.log 2>&1
( timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 ) > $O/run2_bench.json 2>&1
echo finished > $O/run2_done
