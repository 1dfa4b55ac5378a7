cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/gpu_session.sh s49 tests:gru_conv py:gru_bench.py
