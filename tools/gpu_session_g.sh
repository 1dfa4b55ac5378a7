#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r2g
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_pipeline_gpu.py -q -k "corrnet or seam or cfg1 or graph" 2>&1 | tail -3
for i in 1 2; do timeout 120 python tools/corrnet_bench.py 300 2>&1 | tail -1; done
timeout 300 python bench.py --steps 60 --minimal > ${O}_bench.json 2> ${O}_bench.err
python -c "
import json
d=json.loads(open('${O}_bench.json').read()); print('bench', round(d['value'],1), round(d['ms_per_step'],4))"
