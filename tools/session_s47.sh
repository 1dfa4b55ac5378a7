cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(cd tools/ubench/frozen && timeout 300 python bench.py --minimal --steps 20 --warmup 5 --repeats 5 > $GRAFT_REPO_ROOT/gpurun_out/s47_frozen.json 2> $GRAFT_REPO_ROOT/gpurun_out/s47_frozen.err); python tools/bench_digest.py gpurun_out/s47_frozen.json | head -2
bash tools/gpu_session.sh s47 tests:conv+or+pipeline+or+golden+or+teacher bench:--steps+20+--warmup+5+--minimal trace
