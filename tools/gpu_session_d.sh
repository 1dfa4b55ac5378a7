#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r2d
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_teacher_forced_gpu.py tests/test_conv_gpu.py -q -s -k "head or teacher or every_stage or transposed" > ${O}_t1.log 2>&1
ITERMVS_HEAD_FORM=wave timeout 300 python bench.py --steps 50 --minimal > ${O}_bench_wave.json 2> ${O}_bench_wave.err
timeout 300 python bench.py --steps 50 --minimal > ${O}_bench_coop.json 2> ${O}_bench_coop.err
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2d_prof -- python $R/bench.py --steps 20 --warmup 4 --minimal > $R/${O}_prof.log 2>&1
cd $R
grep -E "passed|failed|^FAILED|^ERROR|teacher-forced" ${O}_t1.log | cut -c1-700
for f in wave coop; do python -c "
import json
d=json.loads(open('${O}_bench_$f.json').read()); print('$f', round(d['value'],1), d['ms_per_step'])"; done
python tools/step_timeline.py $(ls gpurun_out/r2d_prof/*/*kernel_trace.csv | head -1) > ${O}_timeline.txt 2>&1; tail -25 ${O}_timeline.txt
