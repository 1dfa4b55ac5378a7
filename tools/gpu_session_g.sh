#!/bin/bash
# other shapes of BASELINE.json's configs (same command, resident inputs): cfg 3 shape, cfg 5 shape, cfg 1 with 16-bit features, batch 2
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r2h
timeout 300 python bench.py --steps 30 --minimal --height 1152 --width 1600 > ${O}_bench_cfg3_shape.json 2> ${O}_cfg3.err
timeout 400 python bench.py --steps 20 --minimal --height 1280 --width 1920 --views 11 --iters 8 > ${O}_bench_cfg5_shape.json 2> ${O}_cfg5.err
timeout 300 python bench.py --steps 60 --minimal --feature-dtype fp16 > ${O}_bench_fp16_features.json 2> ${O}_fp16.err
timeout 300 python bench.py --steps 60 --minimal --feature-dtype bf16 > ${O}_bench_bf16_features.json 2> ${O}_bf16.err
timeout 300 python bench.py --steps 40 --minimal --batch 2 > ${O}_bench_batch2.json 2> ${O}_b2.err
for f in cfg3_shape cfg5_shape fp16_features bf16_features batch2; do python -c "
import json
d=json.loads(open('${O}_bench_$f.json').read()); r=d['roofline']; print('$f', round(d['value'],1), round(d['ms_per_step'],3), 'iter us', round(r['avg_launch_ms']*1e3,1), 'frac', round(r['frac'],3))"; done
