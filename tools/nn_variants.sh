#!/bin/bash
# (GPU box) time brick-kernel build variants: tools/nn_variants.sh "l0_w4 l0_w3 l1_w4"
R=$(cd "$(dirname "$0")/.." && pwd)
for v in $1; do
  echo "== $v"
  PCDHIP_LIB=$R/colmap-pcd_amd/variants/libpcdhip_${v}.so python3 $R/tools/nn_probe.py 1e7 1e6 2 2 2>&1 | grep -E "nn_brick |nn_fallback|kernel sum|sample"
done
