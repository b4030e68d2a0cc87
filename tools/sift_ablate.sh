#!/bin/bash
# timing-only ablations of the SIFT scores kernel: builds colmap-pcd_amd/variants/libpcdhip_sift<k>.so with
# -DPCD_SIFT_ABLATE=<k> (csrc/sift.hip) and runs the 50-image block of tools/sift_probe.py with each.
# usage: tools/sift_ablate.sh build   (here, no GPU needed)      tools/sift_ablate.sh run   (on the GPU box)
set -e
cd "$(dirname "$0")/../colmap-pcd_amd"
VARIANTS="${VARIANTS:-0 1 2 32 64 66 13}"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -I../include"
if [ "$1" = build ]; then
  mkdir -p variants
  for k in $VARIANTS; do
    /opt/rocm/bin/hipcc $FLAGS -DPCD_SIFT_ABLATE=$k ${SIFT_EXTRA} -c csrc/sift.hip -o variants/sift_$k.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o variants/libpcdhip_sift$k.so csrc/common.o csrc/cloud.o csrc/nn.o csrc/assoc.o csrc/ba.o variants/sift_$k.o csrc/proj.o csrc/shards.o
  done
else
  for k in $VARIANTS; do
    echo "PCD_SIFT_ABLATE=$k"
    PROBE_ONLY_BLOCK=1 PCDHIP_LIB=$PWD/variants/libpcdhip_sift$k.so python ../tools/sift_probe.py 2>&1 | grep "batch: 50"
  done
fi
