cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/head_bench.py --all-bf16x3 --lib tools/ubench/variants/libitermvs_head_before.so 2>&1 | grep -v amdgpu.ids
python tools/head_bench.py --all-bf16x3 2>&1 | grep -v amdgpu.ids
bash tools/gpu_session.sh s52 tests:head+or+regress+or+argmax+or+pipeline
