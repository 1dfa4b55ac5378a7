#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r2g
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_pipeline_gpu.py -q -k "corrnet or seam or cfg1 or graph" > ${O}_t1.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR|Error" ${O}_t1.log | cut -c1-400
for rep in 1 2; do for c in 32 28; do
ITERMVS_CORRNET_TR=$c timeout 300 python bench.py --steps 60 --minimal > ${O}_bench_${c}.json 2> ${O}_bench_${c}.err
python -c "
import json
d=json.loads(open('${O}_bench_${c}.json').read()); print('corrnet rows $c', round(d['value'],1), round(d['ms_per_step'],4))"
done; done
