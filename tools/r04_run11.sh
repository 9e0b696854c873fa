#!/bin/bash
# round-4 GPU call 11: 768-point ROWS on the 12 x 64-thread plan (parity + rates); 1024-point Y passes through the DIF-split kernel
# (DFFT_DIF2_MIN=1024, an existing measurement switch) in separate processes
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch_tools.py -q -m gpu -x -k "768 or rows or fft1d or slab or batch" 2>&1 | tail -5 ) > $O/run11_pytest.log 2>&1
SP="512x512x768:fp64:1 768x768x768:fp64:1 768x768x768:fp32:1 512x512x768:fp32:1 1024x768x512:fp64:1"
for rep in 1 2; do timeout 600 python tools/lib_ab.py $SP; done > $O/run11_rows_768_e12.log 2>&1
S2="512x1024x512:fp64:1 512x1024x512:fp32:1 1024x1024x1024:fp32:1 256x1024x1024:fp64:1"
for rep in 1 2; do
  timeout 600 python tools/lib_ab.py $S2
  DFFT_DIF2_MIN=1024 timeout 600 python tools/lib_ab.py $(for s in $S2; do echo $s:DFFT_DIF2_MIN=1024; done)
done > $O/run11_dif2_1024.log 2>&1
echo finished > $O/run11_done
