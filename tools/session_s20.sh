cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/lat_conv_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/s20_lc.log
for v in 1 2 4 7 8 16 24 32 64 128 255; do python tools/lat_conv_bench.py --lib tools/ubench/variants/libitermvs_lc_$v.so --only-fused 2>&1 | grep "one launch" | sed "s/^/ko $v: /" >> gpurun_out/s20_lc.log; done
cat gpurun_out/s20_lc.log
