#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r2g
timeout 900 python -m pytest tests/test_pipeline_gpu.py -q -k "seam or cfg1 or graph" 2>&1 | tail -1
for rep in 1 2; do for c in 0 1; do
ITERMVS_FPN_FORK=$c timeout 300 python bench.py --steps 60 --minimal > ${O}_bench_${c}.json 2> ${O}_bench_${c}.err
python -c "
import json
d=json.loads(open('${O}_bench_${c}.json').read()); print('fork $c', round(d['value'],1), round(d['ms_per_step'],4))"
done; done
