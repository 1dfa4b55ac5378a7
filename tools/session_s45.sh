cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/small_bench.py 2>&1 | grep -v amdgpu.ids | grep -i "upsample\|hidden init" > gpurun_out/s45_small.log; cat gpurun_out/s45_small.log
(cd tools/ubench/frozen && timeout 300 python bench.py --minimal --steps 20 --warmup 5 --repeats 5 > $GRAFT_REPO_ROOT/gpurun_out/s45_frozen.json 2> $GRAFT_REPO_ROOT/gpurun_out/s45_frozen.err); python tools/bench_digest.py gpurun_out/s45_frozen.json | head -2
bash tools/gpu_session.sh s45 tests:conv3x3_conv1x1+or+pipeline+or+golden+or+teacher bench:--steps+20+--warmup+5+--minimal
