cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/gpu_session.sh s42 bench trace pmc bench:--gpus+1+--steps+20+--warmup+5
