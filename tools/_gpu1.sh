mkdir -p gpurun_out/r03b
V="half=DFFT_X_VARIANT=half,full=,fullearly=DFFT_X_VARIANT=fullearly"
timeout 400 python tools/variant_ab.py \
  "512x512x512:fp64:1:3:base=,early=DFFT_X_VARIANT=early,lazy=DFFT_ZY_LAZY=1,both=DFFT_X_VARIANT=early+DFFT_ZY_LAZY=1" \
  "1024x768x512:fp64:1:2:$V" "1024x768x512:fp32:1:2:$V" "1024x768x512:fp64:8:3:$V" \
  "512x512x512:fp32:1:2:base=,early=DFFT_X_VARIANT=early" "1024x1024x1024:fp64:1:1:half=DFFT_X_VARIANT=half,full=" \
  > gpurun_out/r03b/variant_ab.log 2>&1
echo "variant_ab rc=$?" >> gpurun_out/r03b/variant_ab.log
timeout 240 python -m pytest tests/test_gpu_parity.py -q -x -k "one_launch_t0_is or x_pass_prefetch or rotated_exchange" > gpurun_out/r03b/pytest_new.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03b/pytest_new.log
tail -30 gpurun_out/r03b/variant_ab.log; tail -5 gpurun_out/r03b/pytest_new.log
