#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r2g
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_teacher_forced_gpu.py tests/test_pipeline_gpu.py -q -k "corr or every_stage or seam or cfg1" 2>&1 | tail -3
timeout 300 python bench.py --steps 60 --no-transfers --no-cpu-baseline --pipeline-streams 0 > ${O}_bench.json 2> ${O}_bench.err
python -c "
import json
d=json.loads(open('${O}_bench.json').read()); r=d['roofline']; print('bench', round(d['value'],1), round(d['ms_per_step'],4), 'iter us', round(r['avg_launch_ms']*1e3,2), 'init us', round(r['corr_init']['avg_launch_ms']*1e3,2))"
