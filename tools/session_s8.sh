cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/gpu_session.sh s8 py:small_bench.py tests bench:--steps+20+--warmup+5+--no-cpu-baseline+--no-other-configs
