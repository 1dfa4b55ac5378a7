#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r2g
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_pipeline_gpu.py tests/test_teacher_forced_gpu.py -q -k "relu_dot or seam or cfg1 or graph or every_stage" 2>&1 | tail -3
timeout 300 python bench.py --steps 60 --minimal > ${O}_bench.json 2> ${O}_bench.err
python -c "
import json
d=json.loads(open('${O}_bench.json').read()); print('bench', round(d['value'],1), round(d['ms_per_step'],4))"
