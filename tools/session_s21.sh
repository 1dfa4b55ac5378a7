cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/lat_conv_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/s21_lc.log; cat gpurun_out/s21_lc.log
bash tools/gpu_session.sh s21 tests:lateral_conv3x3
