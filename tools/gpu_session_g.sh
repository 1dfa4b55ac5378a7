#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for m in 0 1 2 3 4 8 7 15; do FPN_MODE=$m timeout 120 python tools/fpn_bench.py 1 200; done
