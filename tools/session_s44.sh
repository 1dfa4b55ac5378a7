cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/gpu_session.sh s44 tests bench trace pmc bench:--gpus+1+--steps+20+--warmup+5
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s44_smoke.log 2>&1; tail -1 gpurun_out/s44_smoke.log
