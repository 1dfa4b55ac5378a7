#!/bin/bash
# A/B builds of libitermvs_hip.so for tools/kernel_bench.py --lib (this container; the .so files travel to the GPU box with
# the snapshot, tools/ubench/variants/ is git-ignored):
#   old      corr.hip / corr_common.hpp of a given commit (default: the round-2 head) linked with today's other objects
#   w5/w6/w8 today's corr.hip with __launch_bounds__(256, N) on the two fused correlation kernels
set -e
R=$(cd $(dirname $0)/.. && pwd)
C=$R/itermvs_amd/csrc
V=$R/tools/ubench/variants
OLD=${1:-5c305d0}
rm -rf /tmp/itermvs_variants; mkdir -p $V /tmp/itermvs_variants
make -C $C -j8 > /dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$R/include -Wno-unused-function"
OBJS=$(ls $C/*.o | grep -v "/corr.o")
T=/tmp/itermvs_variants
mkdir -p $T/old
for f in corr.hip corr_common.hpp common.hpp; do git -C $R show $OLD:itermvs_amd/csrc/$f > $T/old/$f; done
/opt/rocm/bin/hipcc $FLAGS -c $T/old/corr.hip -o $T/corr_old.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $T/corr_old.o -o $V/libitermvs_old.so
for w in ${LB_VARIANTS:-}; do
  sed "s/__launch_bounds__(kThreads) corr_iter_kernel/__launch_bounds__(kThreads, $w) corr_iter_kernel/; s/__launch_bounds__(kThreads) corr_init_kernel/__launch_bounds__(kThreads, $w) corr_init_kernel/" $C/corr.hip > $T/corr_w$w.hip
  /opt/rocm/bin/hipcc $FLAGS -I$C -c $T/corr_w$w.hip -o $T/corr_w$w.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $T/corr_w$w.o -o $V/libitermvs_w$w.so
done
# nopx4: today's conv_tile.hip with the four-pixels-per-lane epilogue switched off
sed "s/a.px4 = p->out_layout == 0/a.px4 = false \&\& p->out_layout == 0/" $C/conv_tile.hip > $T/conv_tile_nopx4.hip
/opt/rocm/bin/hipcc $FLAGS -I$C -c $T/conv_tile_nopx4.hip -o $T/conv_tile_nopx4.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/*.o | grep -v "/conv_tile.o") $T/conv_tile_nopx4.o -o $V/libitermvs_nopx4.so
ls -la $V
