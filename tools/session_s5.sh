cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(cd tools/ubench/frozen && timeout 300 python bench.py --minimal --steps 20 --warmup 5 --repeats 5 > $GRAFT_REPO_ROOT/gpurun_out/s5_frozen.json 2> $GRAFT_REPO_ROOT/gpurun_out/s5_frozen.err); python tools/bench_digest.py gpurun_out/s5_frozen.json
{ for i in 1 2; do python tools/step_bench.py --steps 200; python tools/step_bench.py --steps 200 --no-side-branch; done; } 2>&1 | grep "depth map" > gpurun_out/s5_side_branch_ab.txt
cat gpurun_out/s5_side_branch_ab.txt
bash tools/gpu_session.sh s5 tests:pipeline+or+teacher+or+drivers bench:--steps+20+--warmup+5+--no-cpu-baseline+--no-other-configs trace
