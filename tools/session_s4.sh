cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(cd tools/ubench/frozen && timeout 300 python bench.py --minimal --steps 20 --warmup 5 --repeats 5 > $GRAFT_REPO_ROOT/gpurun_out/s4_frozen.json 2> $GRAFT_REPO_ROOT/gpurun_out/s4_frozen.err); python tools/bench_digest.py gpurun_out/s4_frozen.json
V=tools/ubench/variants
{ echo "== head_coop: product build, then every fp32 MFMA group of 4 replaced by 3 bf16 MFMAs (timing only, wrong results)";
  python tools/head_bench.py 200; python tools/head_bench.py 200 --lib $V/libitermvs_head_ko.so;
  echo "== corrnet: product build (float4 staging arm first), then the same knock-out";
  python tools/corrnet_bench.py 200; python tools/corrnet_bench.py 200 --lib $V/libitermvs_corrnet_ko.so; } > gpurun_out/s4_mfma_knockouts.txt 2>&1
grep -v amdgpu.ids gpurun_out/s4_mfma_knockouts.txt
bash tools/gpu_session.sh s4 tests:kernels+or+host+or+conv bench:--steps+20+--warmup+5+--no-cpu-baseline
