#!/bin/bash
# Hardware counters of the path's kernels: separate rocprofv3 --pmc passes (never combined with a trace mode) over a short
# eager bench run (counters are not collected for kernels replayed from a hipGraph), summarised per kernel into
# gpurun_out/<round>_pmc_kernels.json by tools/pmc_summary.py -- copy that file to profiles/ and commit it; bench.py reads
# roofline.traffic from it.  Run on the GPU box from the repo root:   bash tools/pmc_kernels.sh [round-tag]
# PMC_CMD overrides the profiled command (default: the eager bench run), e.g. PMC_CMD="python tools/stem_bench.py 20".
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
CMD=${PMC_CMD:-python $R/bench.py --steps 3 --warmup 2 --repeats 1 --eager --minimal}
OUT=$R/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -- $CMD > $OUT.pass$i.log 2>&1
  echo "pass $i ($set): exit $?"
done
PMC_CMD_USED="$CMD" python $R/tools/pmc_summary.py $OUT $R/gpurun_out/${TAG}_pmc_kernels.json | tail -60
