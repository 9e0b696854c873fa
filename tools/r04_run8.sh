#!/bin/bash
# round-4 GPU call 8: where does t0 of configs 4 / 5 lose?  rocprofv3 kernel trace of rank 0's local work (exchange off) and of the
# single-GPU plan of config 4's shape: per-kernel average durations -> rate of the Z pass and of the Y pass separately
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for spec in 1024x768x512:fp64:1 1024x768x512:fp64:8 2048x2048x1024:fp32:8 512x512x512:fp64:4:DFFT_T0_ONE_LAUNCH=0; do
  tag=$(echo $spec | tr ':x=' '___')
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$tag -- python $R/tools/lib_ab.py $spec > $O/trace_$tag.log 2>&1
  f=$(find $O/trace_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/run8_kernel_stats_$tag.csv
  rm -rf $O/trace_$tag
done
echo finished > $O/run8_done
