cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 --minimal > gpurun_out/s59_a.json 2>/dev/null; python tools/bench_digest.py gpurun_out/s59_a.json | head -1
ITERMVS_SIDE_PAIR=1 python bench.py --steps 20 --warmup 5 --minimal > gpurun_out/s59_b.json 2>/dev/null; python tools/bench_digest.py gpurun_out/s59_b.json | head -1
python bench.py --steps 20 --warmup 5 --minimal > gpurun_out/s59_c.json 2>/dev/null; python tools/bench_digest.py gpurun_out/s59_c.json | head -1
ITERMVS_SIDE_PAIR=1 python bench.py --steps 20 --warmup 5 --minimal > gpurun_out/s59_d.json 2>/dev/null; python tools/bench_digest.py gpurun_out/s59_d.json | head -1
