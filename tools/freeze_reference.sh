#!/bin/bash
# A frozen copy of the package + bench.py + built library under tools/ubench/frozen/ (git-ignored, travels with gpurun):
# the fixed workload every GPU session of a round runs beside the box probe, so that (box probe, frozen-source step time) pairs
# from different boxes calibrate `value_normalised` (itermvs_amd/benchmarks.POOL_MEDIAN, profiles/r06_box_probe.md).
set -e
R=$(cd $(dirname $0)/.. && pwd)
F=$R/tools/ubench/frozen
rm -rf $F; mkdir -p $F
(cd $R && tar -cf - bench.py itermvs_amd/*.py itermvs_amd/libitermvs_hip.so oracle/*.py) | tar -xf - -C $F
mkdir -p $F/profiles
git -C $R rev-parse HEAD > $F/FROZEN_AT
du -sh $F
