#!/bin/bash
# round-4 GPU call 23: the driver's own round-end sequence on the final tree: GPU suite with -x, smoke(), bench line
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 ) > $O/run23_pytest_x.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $O/run23_smoke.log 2>&1
( timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 ) > $O/run23_bench.json 2>&1
echo finished > $O/run23_done
