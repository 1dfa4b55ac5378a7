#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r2b
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "backward or corr_iter_forms or ragged" -s > ${O}_t1.log 2>&1
timeout 1500 python -m pytest tests/test_teacher_forced_gpu.py "tests/test_pipeline_gpu.py::test_small_pipeline_every_seam" \
   tests/test_train_gpu.py tests/test_drivers_gpu.py -q -s > ${O}_t2.log 2>&1
timeout 120 python tools/transfer_overlap.py > ${O}_overlap.log 2>&1
for impl in 1 3 13 2; do
  ITERMVS_CORR_ITER_IMPL=$impl timeout 300 python bench.py --steps 50 --minimal > ${O}_bench_impl$impl.json 2> ${O}_bench_impl$impl.err
done
tail -3 ${O}_t1.log; grep -E "teacher-forced|e2e_small|train cfg4|train regress|train noregress|passed|failed|^FAILED|AssertionError" ${O}_t2.log | cut -c1-2500
cat ${O}_overlap.log | tail -5
for impl in 1 3 13 2; do python -c "
import json,sys
d=json.loads(open('${O}_bench_impl$impl.json').read()); print($impl, d['value'], d['roofline']['avg_launch_ms'], d['roofline']['corr_init']['avg_launch_ms'])"; done
