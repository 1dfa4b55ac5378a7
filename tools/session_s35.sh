cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(cd tools/ubench/frozen && timeout 300 python bench.py --minimal --steps 20 --warmup 5 --repeats 5 > $GRAFT_REPO_ROOT/gpurun_out/s35_frozen.json 2> $GRAFT_REPO_ROOT/gpurun_out/s35_frozen.err); python tools/bench_digest.py gpurun_out/s35_frozen.json | head -2
(python tools/res_chain_bench.py; python tools/lat_conv_bench.py) 2>&1 | grep -v amdgpu.ids > gpurun_out/s35_rc.log; cat gpurun_out/s35_rc.log
bash tools/gpu_session.sh s35 bench:--steps+20+--warmup+5+--minimal tests:conv+or+pipeline trace pmc
