cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/gpu_session.sh s60 tests:pvw_aggregate+or+pipeline+or+engine+or+drop_in+or+init bench:--steps+20+--warmup+5+--minimal trace
