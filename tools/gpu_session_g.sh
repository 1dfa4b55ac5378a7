#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r2g
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_pipeline_gpu.py tests/test_teacher_forced_gpu.py -q -k "up2 or interp or lateral or seam or cfg1 or graph or every_stage or conv2d" 2>&1 | tail -3
timeout 300 python bench.py --steps 60 --minimal > ${O}_bench.json 2> ${O}_bench.err
python -c "
import json
d=json.loads(open('${O}_bench.json').read()); print('bench', round(d['value'],1), round(d['ms_per_step'],4))"
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/r2g_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2g_prof -- python $R/bench.py --steps 20 --warmup 4 --minimal > $R/${O}_prof.log 2>&1
cd $R
python tools/step_timeline.py $(ls -t gpurun_out/r2g_prof/*/*kernel_trace.csv | head -1) > ${O}_timeline.txt 2>&1; grep -E "busy|conv_mfma_kernel<3" ${O}_timeline.txt | head -8
