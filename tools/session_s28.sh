cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(python tools/head_bench.py; python tools/head_bench.py --w2-bf16x3) 2>&1 | grep -v amdgpu.ids > gpurun_out/s28_head.log; cat gpurun_out/s28_head.log
bash tools/gpu_session.sh s28 tests:head_fused+or+test_small_pipeline+or+test_cfg1 bench:--steps+20+--warmup+5+--minimal
