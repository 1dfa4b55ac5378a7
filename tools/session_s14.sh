cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/res_chain_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/s14_rc.log; cat gpurun_out/s14_rc.log
bash tools/gpu_session.sh s14 tests:res_chain
