#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r2e
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_kernels_gpu.py tests/test_teacher_forced_gpu.py tests/test_pipeline_gpu.py -q -s -k "corrnet or relu_dot or head or every_stage or seam or cfg1 or graph or pvw" > ${O}_t1.log 2>&1
ITERMVS_CORRNET=layers timeout 300 python bench.py --steps 50 --minimal > ${O}_bench_layers.json 2> ${O}_bench_layers.err
timeout 300 python bench.py --steps 50 --minimal > ${O}_bench_fused.json 2> ${O}_bench_fused.err
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2e_prof -- python $R/bench.py --steps 20 --warmup 4 --minimal > $R/${O}_prof.log 2>&1
cd $R
grep -E "passed|failed|^FAILED|^ERROR|Error" ${O}_t1.log | cut -c1-400
for f in layers fused; do python -c "
import json
d=json.loads(open('${O}_bench_$f.json').read()); print('$f', round(d['value'],1), d['ms_per_step'])"; done
python tools/step_timeline.py $(ls gpurun_out/r2e_prof/*/*kernel_trace.csv | head -1) > ${O}_timeline.txt 2>&1; tail -16 ${O}_timeline.txt; grep -E "corrnet|head_coop" ${O}_timeline.txt | head -12
