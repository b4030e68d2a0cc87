#!/bin/bash
# Build variants of libpcdhip.so with different brick-kernel knobs (HERE, in the container: hipcc
# cross-compiles) and print the commands to time them on the GPU box.
#   tools/nn_tune.sh build  "256:4 192:5 192:6 128:6 128:8"     -> colmap-pcd_amd/variants/libpcdhip_<tile>_<waves>.so
#   (GPU box) tools/nn_tune.sh run  "256:4 192:5 ..."            -> times each with tools/nn_probe.py
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/colmap-pcd_amd
mode=$1; shift
vars=${1:-"256:4 192:5 192:6 128:6 128:8"}
if [ "$mode" = build ]; then
  mkdir -p $P/variants
  for v in $vars; do
    t=${v%%:*}; w=${v##*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -I$R/include \
      -DPCD_KTILE=$t -DPCD_BRICK_MINWAVES=$w -c $P/csrc/nn.hip -o $P/variants/nn_${t}_${w}.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $P/variants/libpcdhip_${t}_${w}.so \
      $P/csrc/common.o $P/csrc/cloud.o $P/variants/nn_${t}_${w}.o $P/csrc/assoc.o $P/csrc/ba.o
    echo "built $v"
  done
else
  for v in $vars; do
    t=${v%%:*}; w=${v##*:}
    echo "== tile $t, $w waves/SIMD"
    PCDHIP_LIB=$P/variants/libpcdhip_${t}_${w}.so python3 $R/tools/nn_probe.py 1e7 1e6 2 2 2>&1 | grep -E "nn_brick |nn_fallback|kernel sum|sample"
  done
fi
