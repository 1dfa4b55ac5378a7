cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(cd tools/ubench/frozen && timeout 600 python bench.py --steps 20 --warmup 5 --minimal > $GRAFT_REPO_ROOT/gpurun_out/s62_frozen.json 2>/dev/null); python tools/bench_digest.py gpurun_out/s62_frozen.json
bash tools/gpu_session.sh s62 tests bench trace pmc bench:--gpus+1+--steps+20+--warmup+5
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/s62_smoke.log 2>&1; tail -2 gpurun_out/s62_smoke.log
