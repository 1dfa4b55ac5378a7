#!/bin/bash
# One GPU-box session (run through gpurun from the repo root), parameterised by the steps wanted:
#     bash tools/gpu_session.sh <tag> <step> [<step> ...]
# steps:  tests        full `pytest -m gpu`                               -> gpurun_out/<tag>_tests.log
#         tests:<k>    `pytest -m gpu -k <k>` (use + for spaces)          -> gpurun_out/<tag>_tests_<n>.log
#         bench        the default bench line                             -> gpurun_out/<tag>_bench.json
#         bench:<args> bench.py with extra flags (use + for spaces)       -> gpurun_out/<tag>_bench_<n>.json
#         trace        rocprofv3 --kernel-trace --stats of a minimal bench run + per-step timeline
#         pmc          counter passes (tools/pmc_kernels.sh) + summary json
#         py:<script+args>   any tools/*.py script                        -> gpurun_out/<tag>_py_<n>.log
#         sh:<command>       any shell command (use + for spaces)         -> gpurun_out/<tag>_sh_<n>.log
#         ubench:<name>      build and run tools/ubench/<name>.hip        -> gpurun_out/<tag>_ubench_<name>.log
# Every step runs under its own `timeout`; outputs land in gpurun_out/ (copy what should be judged into profiles/).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export MIOPEN_LOG_LEVEL=3
mkdir -p gpurun_out
TAG=$1; shift
O=gpurun_out/$TAG
n=0
for step in "$@"; do
  n=$((n+1))
  arg=$(echo "${step#*:}" | tr '+' ' ')
  case "$step" in
    tests)     timeout 1500 python -m pytest tests -m gpu -q -s --timeout 600 > ${O}_tests.log 2>&1; tail -3 ${O}_tests.log; grep -E "^FAILED|^ERROR" ${O}_tests.log | head -20 ;;
    tests:*)   timeout 1500 python -m pytest tests -m gpu -q -s --timeout 600 -k "$arg" > ${O}_tests_$n.log 2>&1; tail -3 ${O}_tests_$n.log; grep -E "^FAILED|^ERROR" ${O}_tests_$n.log | head -20 ;;
    bench)     timeout 900 python bench.py > ${O}_bench.json 2> ${O}_bench.err; python tools/bench_digest.py ${O}_bench.json ;;
    bench:*)   timeout 900 python bench.py $arg > ${O}_bench_$n.json 2> ${O}_bench_$n.err; python tools/bench_digest.py ${O}_bench_$n.json ;;
    trace)     (cd /tmp && export TMPDIR=/tmp && rm -rf $R/${O}_prof && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/${O}_prof -- python $R/bench.py --steps 30 --warmup 6 --repeats 1 --minimal > $R/${O}_prof.log 2>&1)
               python tools/step_timeline.py $(ls ${O}_prof/*/*kernel_trace.csv | head -1) > ${O}_timeline.txt 2>&1
               cp $(ls ${O}_prof/*/*kernel_stats.csv | head -1) ${O}_kernel_stats.csv; tail -16 ${O}_timeline.txt ;;
    pmc)       bash tools/pmc_kernels.sh $TAG > ${O}_pmc.log 2>&1; tail -5 ${O}_pmc.log ;;
    py:*)      timeout 900 python tools/$arg > ${O}_py_$n.log 2>&1; tail -40 ${O}_py_$n.log ;;
    sh:*)      (timeout 900 bash -c "$arg") > ${O}_sh_$n.log 2>&1; tail -40 ${O}_sh_$n.log ;;
    ubench:*)  (cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $arg $arg.hip 2>&1 | tail -3 && timeout 300 ./$arg) > ${O}_ubench_$arg.log 2>&1; cat ${O}_ubench_$arg.log ;;
    *)         echo "unknown step $step" ;;
  esac
done
