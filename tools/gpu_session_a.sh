#!/bin/bash
# one GPU-box session: new corr_iter form (parity + micro-benchmark), teacher-forced parity, bench A/B, counters
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r2a
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "corr_iter or ragged or compose_proj" -s 2>&1 | tail -25 > ${O}_t1.log
timeout 400 python tools/kernel_bench.py > ${O}_kbench.log 2>&1
timeout 1200 python -m pytest tests/test_teacher_forced_gpu.py "tests/test_pipeline_gpu.py::test_small_pipeline_every_seam" \
   "tests/test_pipeline_gpu.py::test_nan_projection_raises_like_the_reference_and_clears" tests/test_train_gpu.py tests/test_fusion.py \
   tests/test_drivers_gpu.py -q -s 2>&1 | tail -80 > ${O}_t2.log
ITERMVS_CORR_ITER_IMPL=lane timeout 600 python bench.py --steps 50 --no-cpu-baseline > ${O}_bench_lane.json 2> ${O}_bench_lane.err
timeout 600 python bench.py --steps 50 > ${O}_bench_views.json 2> ${O}_bench_views.err
bash tools/pmc_kernels.sh r02a > ${O}_pmc.log 2>&1
ITERMVS_CORR_ITER_IMPL=lane bash tools/pmc_kernels.sh r02a_lane > ${O}_pmc_lane.log 2>&1
tail -5 ${O}_t1.log ${O}_t2.log; cat ${O}_kbench.log; cat ${O}_bench_views.json | cut -c1-1500
