#!/bin/bash
# A/B builds of libitermvs_hip.so for tools/kernel_bench.py --lib (this container; the .so files travel to the GPU box with
# the snapshot, tools/ubench/variants/ is git-ignored):
#   tw8 / tw32   the fused correlation kernels with other pixel-tile shapes
#   LB_VARIANTS="5 6" adds corr.hip with __launch_bounds__(256, N) on the two fused correlation kernels
set -e
R=$(cd $(dirname $0)/.. && pwd)
C=$R/itermvs_amd/csrc
V=$R/tools/ubench/variants
rm -rf /tmp/itermvs_variants; mkdir -p $V /tmp/itermvs_variants
make -C $C -j8 > /dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$R/include -Wno-unused-function"
OBJS=$(ls $C/*.o | grep -v "/corr.o")
T=/tmp/itermvs_variants
for w in ${LB_VARIANTS:-}; do
  sed "s/__launch_bounds__(kThreads) corr_iter_kernel/__launch_bounds__(kThreads, $w) corr_iter_kernel/; s/__launch_bounds__(kThreads) corr_init_kernel/__launch_bounds__(kThreads, $w) corr_init_kernel/" $C/corr.hip > $T/corr_w$w.hip
  /opt/rocm/bin/hipcc $FLAGS -I$C -c $T/corr_w$w.hip -o $T/corr_w$w.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $T/corr_w$w.o -o $V/libitermvs_w$w.so
done
# tw8 / tw32: the fused correlation kernels with 8 x 4 / 32 x 1 pixel tiles (default: 16 x 2)
for tw in ${TW_VARIANTS-8 32}; do
  /opt/rocm/bin/hipcc $FLAGS -DITERMVS_CORR_TW=$tw -I$C -c $C/corr.hip -o $T/corr_tw$tw.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $T/corr_tw$tw.o -o $V/libitermvs_tw$tw.so
done
# tuning: the whole library with -DITERMVS_TUNING (reads the ITERMVS_* overrides)
if [ "${TUNING_LIB:-0}" = "1" ]; then
  mkdir -p $T/tuning
  for f in $C/*.hip; do b=$(basename $f .hip); /opt/rocm/bin/hipcc $FLAGS -DITERMVS_TUNING -I$C -c $f -o $T/tuning/$b.o & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $T/tuning/*.o -o $V/libitermvs_tuning.so
fi

# corr: corr.hip with other -D switches, e.g. CORR_VARIANTS="pair:-DITERMVS_FT16_PAIR=1 pair_lb2:-DITERMVS_FT16_PAIR=1,-DITERMVS_CORR_WAVES=4"
for spec in ${CORR_VARIANTS:-}; do
  name=${spec%%:*}; defs=$(echo "${spec#*:}" | tr ',' ' ')
  /opt/rocm/bin/hipcc $FLAGS $defs -I$C -c $C/corr.hip -o $T/corr_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $T/corr_$name.o -o $V/libitermvs_corr_$name.so
done
# tile: conv_tile.hip with other -D switches, e.g. TILE_VARIANTS="dma:-DITERMVS_TILE_DMA"
for spec in ${TILE_VARIANTS:-}; do
  name=${spec%%:*}; defs=$(echo "${spec#*:}" | tr ',' ' ')
  /opt/rocm/bin/hipcc $FLAGS $defs -I$C -c $C/conv_tile.hip -o $T/conv_tile_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/*.o | grep -v "/conv_tile.o") $T/conv_tile_$name.o -o $V/libitermvs_tile_$name.so
done
# bwd: corr_bwd.hip with other -D switches, e.g. BWD_VARIANTS="qt:-DITERMVS_BWD_QT=1"
for spec in ${BWD_VARIANTS:-}; do
  name=${spec%%:*}; defs=$(echo "${spec#*:}" | tr ',' ' ')
  /opt/rocm/bin/hipcc $FLAGS $defs -I$C -c $C/corr_bwd.hip -o $T/corr_bwd_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/*.o | grep -v "/corr_bwd.o") $T/corr_bwd_$name.o -o $V/libitermvs_bwd_$name.so
done
ls $V
# bwd diagnostics (sed-made copies of corr_bwd.hip; results are WRONG, timing only):
#   BWD_DIAG="noatomic noload noalu wg6"   scatter without the atomics / without the tap loads / with a trivial footprint
#                                          instead of the projection / 6 workgroups per CU
for d in ${BWD_DIAG:-}; do
  case $d in
    noatomic) sed 's|unsafeAtomicAdd(gb + (st.o\[tp\] + 16u \* blk), \(.*\));|{ float keep_ = \1; asm volatile("" :: "v"(keep_), "v"(st.o[tp])); }|' $C/corr_bwd.hip > $T/corr_bwd_$d.hip ;;
    noload)   sed 's|st.tap\[4 \* blk + tp\] = ld_feat<FT>(fb, st.o\[tp\] + 16u \* blk);|st.tap[4 * blk + tp] = __int_as_float((int)st.o[tp]);|' $C/corr_bwd.hip > $T/corr_bwd_$d.hip ;;
    noalu)    sed 's|^            project_fast(g, rc, m, rx, ry, rz, d, ix, iy);$|            ix = xs + 1.7f * (float)n; iy = ys + 0.3f; (void)d; (void)rx;|' $C/corr_bwd.hip > $T/corr_bwd_$d.hip ;;
    wg[0-9])  sed "s|__launch_bounds__(kThreads) corr_bwd_kernel|__launch_bounds__(kThreads, ${d#wg}) corr_bwd_kernel|" $C/corr_bwd.hip > $T/corr_bwd_$d.hip ;;
  esac
  cmp -s $C/corr_bwd.hip $T/corr_bwd_$d.hip && { echo "variant $d: sed matched nothing"; exit 1; }
  /opt/rocm/bin/hipcc $FLAGS -I$C -c $T/corr_bwd_$d.hip -o $T/corr_bwd_$d.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/*.o | grep -v "/corr_bwd.o") $T/corr_bwd_$d.o -o $V/libitermvs_bwd_$d.so
done
ls $V
# level masks: corr_iter_kernel with only the levels of the mask doing work (timing only), e.g. LEVEL_MASKS="1 2 4"
for mk in ${LEVEL_MASKS-}; do
  sed "s|    const int lvl = blockIdx.y;|    const int lvl = blockIdx.y; if (!(($mk >> lvl) \& 1)) return;|" $C/corr.hip > $T/corr_lm$mk.hip
  cmp -s $C/corr.hip $T/corr_lm$mk.hip && { echo "level mask: sed matched nothing"; exit 1; }
  /opt/rocm/bin/hipcc $FLAGS -I$C -c $T/corr_lm$mk.hip -o $T/corr_lm$mk.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $T/corr_lm$mk.o -o $V/libitermvs_lm$mk.so
done
ls $V
