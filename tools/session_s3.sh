cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(cd tools/ubench/frozen && timeout 300 python bench.py --minimal --steps 20 --warmup 5 --repeats 5 > $GRAFT_REPO_ROOT/gpurun_out/s3_frozen.json 2> $GRAFT_REPO_ROOT/gpurun_out/s3_frozen.err); python tools/bench_digest.py gpurun_out/s3_frozen.json
T=tools/ubench/variants/libitermvs_tuning.so
for shape in "--height 1280 --width 1920 --views 11" "--height 1152 --width 1600 --views 5" "--height 512 --width 640 --views 5"; do
  for dt in fp32 fp16; do
    for rows in 0 -1 1 2 4 8 16; do
      echo "== $shape $dt band_rows=$rows"
      ITERMVS_ITER_BAND_ROWS=$rows timeout 300 python tools/kernel_bench.py $shape --dtype $dt --lib $T 2>&1 | grep "corr_iter"
    done
  done
done > gpurun_out/s3_band_sweep.txt 2>&1
cat gpurun_out/s3_band_sweep.txt
bash tools/gpu_session.sh s3 py:small_bench.py tests bench:--steps+20+--warmup+5+--no-other-configs+--no-cpu-baseline trace
