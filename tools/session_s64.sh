cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline > gpurun_out/s64_bench.json 2> gpurun_out/s64_bench.err; python tools/bench_digest.py gpurun_out/s64_bench.json | head -8
bash tools/gpu_session.sh s64 tests:gru_conv+or+bench+or+drivers
