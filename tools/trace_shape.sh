R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r04i
(cd /tmp && export TMPDIR=/tmp && rm -rf $R/${O}_prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/${O}_prof -- python $R/bench.py --steps 6 --warmup 3 --repeats 1 --minimal --height 1280 --width 1920 --views 11 --iters 8 --feature-dtype fp16 > $R/${O}_prof.log 2>&1)
python tools/step_timeline.py $(ls ${O}_prof/*/*kernel_trace.csv | head -1) > ${O}_timeline_cfg5.txt 2>&1
tail -30 ${O}_timeline_cfg5.txt
