cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/gpu_session.sh s32 tests bench trace pmc
