cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/small_bench.py 2>&1 | grep -v amdgpu.ids | grep "itermvs_conv3x3_conv1x1"
bash tools/gpu_session.sh s53 tests:conv3x3_conv1x1+or+pipeline+or+head bench:--steps+20+--warmup+5+--minimal
