#!/bin/bash
# full GPU suite, the bench line, kernel trace + step timeline, counters
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r2f
timeout 1500 python -m pytest tests -m gpu -q > ${O}_tests.log 2>&1
timeout 900 python bench.py > ${O}_bench.json 2> ${O}_bench.err
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/r2f_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2f_prof -- python $R/bench.py --steps 30 --warmup 6 --minimal > $R/${O}_prof.log 2>&1
cd $R
python tools/step_timeline.py $(ls gpurun_out/r2f_prof/*/*kernel_trace.csv | head -1) > ${O}_timeline.txt 2>&1
cp $(ls gpurun_out/r2f_prof/*/*kernel_stats.csv | head -1) ${O}_kernel_stats.csv
bash tools/pmc_kernels.sh r02 > ${O}_pmc.log 2>&1
tail -3 ${O}_tests.log; grep -E "^FAILED|^ERROR" ${O}_tests.log | head
python -c "
import json
d=json.loads(open('${O}_bench.json').read()); r=d['roofline']
print('value', round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'iter us', round(r['avg_launch_ms']*1e3,1), 'frac', round(r['frac'],3), 'traffic', r.get('traffic'), 'init us', round(r['corr_init']['avg_launch_ms']*1e3,1))
print('pipelined', d['pipelined']['value'], 'transfers', d['with_transfers']['value'], 'u8', d['with_transfers']['uint8_images']['value'], 'conv frac', d['roofline_conv']['frac'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
tail -14 ${O}_timeline.txt
