cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PMC_CMD="python $GRAFT_REPO_ROOT/tools/lat_conv_bench.py --eager 10" bash tools/pmc_kernels.sh s22 > gpurun_out/s22_pmc.log 2>&1
grep -A40 lat_conv gpurun_out/s22_pmc_kernels.json | head -80
