#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r2c
timeout 1500 python -m pytest tests -m gpu -q -s > ${O}_tests.log 2>&1
ITERMVS_TRACE_TRANSFERS=1 timeout 600 python bench.py --steps 50 > ${O}_bench_fp32.json 2> ${O}_bench_fp32.err
timeout 300 python bench.py --steps 50 --minimal --feature-dtype bf16 > ${O}_bench_bf16.json 2> ${O}_bench_bf16.err
timeout 300 python bench.py --steps 50 --minimal --feature-dtype fp16 > ${O}_bench_fp16.json 2> ${O}_bench_fp16.err
grep -E "passed|failed|^FAILED|^ERROR|feature storage|compose_proj tap|train cfg4" ${O}_tests.log | cut -c1-600
grep "transfers step" ${O}_bench_fp32.err
for f in fp32 bf16 fp16; do python -c "
import json
d=json.loads(open('${O}_bench_$f.json').read()); r=d['roofline']; print('$f', round(d['value'],1), 'iter %.1f us frac %.3f' % (r['avg_launch_ms']*1e3, r['frac']), 'init %.1f us' % (r['corr_init']['avg_launch_ms']*1e3), d.get('with_transfers') and round(d['with_transfers']['value'],1))"; done
