#!/bin/bash
# round-4 GPU call 4: the three cases that failed in call 3 with their full reports; plane-padding sweep of the hand-over buffer for
# the long X axes (1024- and 2048-point X passes)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "plan_tune_keeps or (with_exchange_is_bit_identical and N3)" 2>&1 | grep -v "^    \|^$" | head -150 ) > $O/run4_pytest.log 2>&1
( timeout 900 python tools/variant_ab.py \
   "1024x768x512:fp64:1:2:p3=,p1=DFFT_PAD_PLANE=1,p5=DFFT_PAD_PLANE=5,p9=DFFT_PAD_PLANE=9,p17=DFFT_PAD_PLANE=17,p33=DFFT_PAD_PLANE=33,r1=DFFT_PAD_ROW=1+DFFT_PAD_PLANE=0" \
   "2048x1024x512:fp32:1:2:p3=,p1=DFFT_PAD_PLANE=1,p5=DFFT_PAD_PLANE=5,p9=DFFT_PAD_PLANE=9,p17=DFFT_PAD_PLANE=17,p33=DFFT_PAD_PLANE=33" ) > $O/run4_pad_sweep.log 2>&1
echo finished > $O/run4_done
