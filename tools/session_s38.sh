cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/corrnet_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/s38_cn.log; cat gpurun_out/s38_cn.log
bash tools/gpu_session.sh s38 tests:corrnet
