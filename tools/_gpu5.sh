O=gpurun_out/r03b; mkdir -p $O
V="twolaunch=DFFT_T0_ONE_LAUNCH=0,lazy=,lazy_wgs2=DFFT_ZY_WGS=2,lazy_wgs3=DFFT_ZY_WGS=3,eager=DFFT_ZY_LAZY=0,eager_wgs2=DFFT_ZY_LAZY=0+DFFT_ZY_WGS=2"
timeout 300 python tools/variant_ab.py "256x256x256:fp64:1:3:$V" "512x256x256:fp64:1:2:$V" "256x256x512:fp64:1:2:$V" "256x512x512:fp64:1:2:twolaunch=DFFT_T0_ONE_LAUNCH=0,lazy=,eager=DFFT_ZY_LAZY=0" > $O/variant_ab_256.log 2>&1
echo "rc=$?" >> $O/variant_ab_256.log
cut -c1-175 $O/variant_ab_256.log
