#!/bin/bash
# A/B of the 2048-point column kernels: DIF-split full-line tiles (default) against the half-line tiles (DFFT_NO_DIF2=1).
OUT=${1:-gpurun_out/r02/ab_dif2.log}
mkdir -p "$(dirname "$OUT")"
for v in 0 1 0 1; do
    echo "== DFFT_NO_DIF2=$v" | tee -a "$OUT"
    DFFT_NO_DIF2=$v python tools/long_axis_bench.py 2>/dev/null | grep -E "^(rows|cols),2048|^[0-9]+x" | tee -a "$OUT"
done
