cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(python tools/res_chain_bench.py; python tools/lat_conv_bench.py) 2>&1 | grep -v amdgpu.ids > gpurun_out/s30_rc.log; cat gpurun_out/s30_rc.log
bash tools/gpu_session.sh s30 tests:res_chain+or+lateral_conv bench:--steps+20+--warmup+5+--minimal trace
