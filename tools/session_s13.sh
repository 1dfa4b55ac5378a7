cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/res_chain_bench.py > gpurun_out/s13_rc.log 2>&1
for v in mfma fetch ds store mem all; do python tools/res_chain_bench.py --lib tools/ubench/variants/libitermvs_rc_$v.so --only-chain 2>&1 | grep -v amdgpu.ids | sed "s/^/ko $v: /" >> gpurun_out/s13_rc.log; done
cat gpurun_out/s13_rc.log
