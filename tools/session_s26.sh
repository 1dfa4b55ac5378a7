cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/res_chain_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/s26_rc.log; cat gpurun_out/s26_rc.log
bash tools/gpu_session.sh s26 tests:res_chain
