cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(cd tools/ubench/frozen && timeout 600 python bench.py --steps 20 --warmup 5 --minimal > $GRAFT_REPO_ROOT/gpurun_out/s65_frozen.json 2>/dev/null); python tools/bench_digest.py gpurun_out/s65_frozen.json | head -1
bash tools/gpu_session.sh s65 bench bench:--gpus+1+--steps+20+--warmup+5 tests
