cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/corrnet_bench.py 200 --lib tools/ubench/variants/libitermvs_before.so 2>&1 | grep -v amdgpu.ids | tail -4
python tools/corrnet_bench.py 200 2>&1 | grep -v amdgpu.ids | tail -4
bash tools/gpu_session.sh s61 tests:corrnet
