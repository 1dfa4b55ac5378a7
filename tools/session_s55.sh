cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(cd tools/ubench/frozen && timeout 600 python bench.py --steps 20 --warmup 5 --minimal > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_frozen.json 2>/dev/null); python tools/bench_digest.py gpurun_out/${TAG}_frozen.json | head -1
bash tools/gpu_session.sh $TAG bench:--steps+20+--warmup+5+--minimal | head -1
