cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/gpu_session.sh s10 tests:corrnet py:corrnet_bench.py bench:--steps+20+--warmup+5+--no-cpu-baseline+--no-other-configs tests
