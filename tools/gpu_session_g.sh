#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for m in 0 4 0 4; do echo -n "mode=$m "; STEM_MODE=$m timeout 120 python tools/stem_bench.py 300; done
