cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/lat_conv_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/s19_lc.log; cat gpurun_out/s19_lc.log
(cd tools/ubench/frozen && timeout 300 python bench.py --minimal --steps 20 --warmup 5 --repeats 5 > $GRAFT_REPO_ROOT/gpurun_out/s19_frozen.json 2> $GRAFT_REPO_ROOT/gpurun_out/s19_frozen.err); python tools/bench_digest.py gpurun_out/s19_frozen.json | head -2
bash tools/gpu_session.sh s19 tests:lateral_conv3x3 bench:--steps+20+--warmup+5+--minimal tests:test_e2e+or+test_cfg1+or+test_small_pipeline+or+golden
