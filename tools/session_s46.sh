cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(python tools/head_bench.py; python tools/head_bench.py --w2-bf16x3; python tools/head_bench.py --all-bf16x3) 2>&1 | grep -v amdgpu.ids > gpurun_out/s46_head.log; cat gpurun_out/s46_head.log
(cd tools/ubench/frozen && timeout 300 python bench.py --minimal --steps 20 --warmup 5 --repeats 5 > $GRAFT_REPO_ROOT/gpurun_out/s46_frozen.json 2> $GRAFT_REPO_ROOT/gpurun_out/s46_frozen.err); python tools/bench_digest.py gpurun_out/s46_frozen.json | head -2
bash tools/gpu_session.sh s46 tests:head+or+pipeline+or+golden+or+teacher bench:--steps+20+--warmup+5+--minimal
