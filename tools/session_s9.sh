cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/gpu_session.sh s9 tests:drivers+or+host bench:--steps+20+--warmup+5+--no-cpu-baseline+--no-other-configs bench:--steps+20+--warmup+5+--minimal
