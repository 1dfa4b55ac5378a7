cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/res_chain_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/s15_rc.log; cat gpurun_out/s15_rc.log
(cd tools/ubench/frozen && timeout 300 python bench.py --minimal --steps 20 --warmup 5 --repeats 5 > $GRAFT_REPO_ROOT/gpurun_out/s15_frozen.json 2> $GRAFT_REPO_ROOT/gpurun_out/s15_frozen.err); python tools/bench_digest.py gpurun_out/s15_frozen.json | head -2
bash tools/gpu_session.sh s15 tests:res_chain+or+stem bench:--steps+20+--warmup+5+--minimal tests trace
