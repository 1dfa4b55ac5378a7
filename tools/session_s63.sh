cd $GRAFT_REPO_ROOT
python tools/small_bench.py --lib tools/ubench/variants/libitermvs_before.so 2>&1 | grep -v amdgpu.ids | grep "conv3x3_conv1x1\|depth head"
python tools/small_bench.py 2>&1 | grep -v amdgpu.ids | grep "conv3x3_conv1x1\|depth head"
python tools/small_bench.py --lib tools/ubench/variants/libitermvs_before.so 2>&1 | grep -v amdgpu.ids | grep "conv3x3_conv1x1\|depth head"
python tools/small_bench.py 2>&1 | grep -v amdgpu.ids | grep "conv3x3_conv1x1\|depth head"
