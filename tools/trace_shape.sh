#!/bin/bash
# rocprofv3 kernel trace + per-step timeline of bench.py at another shape / setting (GPU box, from the repo root):
#     bash tools/trace_shape.sh <tag> [bench.py flags ...]
# e.g. bash tools/trace_shape.sh cfg5_fp16 --height 1280 --width 1920 --views 11 --iters 8 --feature-dtype fp16
# -> gpurun_out/<tag>_timeline.txt (one depth map kernel by kernel + the top kernels), gpurun_out/<tag>_prof/ (raw csv)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
TAG=${1:-shape}; shift
O=gpurun_out/$TAG
mkdir -p gpurun_out
(cd /tmp && export TMPDIR=/tmp && rm -rf $R/${O}_prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/${O}_prof -- \
    python $R/bench.py --steps 6 --warmup 3 --repeats 1 --minimal "$@" > $R/${O}_prof.log 2>&1)
python tools/step_timeline.py $(ls ${O}_prof/*/*kernel_trace.csv | head -1) > ${O}_timeline.txt 2>&1
tail -30 ${O}_timeline.txt
