cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(cd tools/ubench/frozen && timeout 300 python bench.py --minimal --steps 20 --warmup 5 --repeats 5 > $GRAFT_REPO_ROOT/gpurun_out/s6_frozen.json 2> $GRAFT_REPO_ROOT/gpurun_out/s6_frozen.err); python tools/bench_digest.py gpurun_out/s6_frozen.json
bash tools/gpu_session.sh s6 tests bench:--steps+20+--warmup+5+--no-cpu-baseline+--no-other-configs trace
