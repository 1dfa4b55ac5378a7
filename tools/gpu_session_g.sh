#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for w in 1 2 3 4 5 2 3; do ITERMVS_HEAD_WGS=$w timeout 120 python tools/head_bench.py 300 2>&1 | tail -1; done
