cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(cd tools/ubench/frozen && timeout 300 python bench.py --minimal --steps 20 --warmup 5 --repeats 5 > $GRAFT_REPO_ROOT/gpurun_out/s2_frozen.json 2> $GRAFT_REPO_ROOT/gpurun_out/s2_frozen.err); python tools/bench_digest.py gpurun_out/s2_frozen.json
bash tools/gpu_session.sh s2 py:small_bench.py py:corrnet_bench.py py:head_bench.py tests bench:--steps+20+--warmup+5+--no-other-configs+--no-cpu-baseline trace
