#!/bin/bash
# HBM traffic of itermvs_corr_iter: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over a short bench run,
# (--eager: counters are not collected for kernels replayed from a hipGraph)
# summarised into profiles/r01_corr_iter_pmc.json by tools/pmc_summary.py.  Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/pmc_corr
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_corr/$c -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --pipeline-streams 0 --eager > /dev/null 2>&1
done
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_corr corr_iter_kernel $R/gpurun_out/r01_corr_iter_pmc.json | tail -12
