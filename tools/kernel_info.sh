#!/bin/bash
# usage: tools/kernel_info.sh csrc/nn.hip  -> VGPR/SGPR/LDS/scratch of each pcd kernel + FMA count in the ISA
set -e
src=$(realpath "$1"); out=/tmp/isa_$$; mkdir -p $out; cd $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -I/root/repo/include \
  -S --cuda-device-only "$src" -o k.s 2>/dev/null
python3 - <<PY
import re
s=open('k.s').read()
for m in re.finditer(r'\.name:\s+(_ZN3pcd\S+).*?(?=\n  - \.a|\namdhsa\.target|\Z)', s, re.S):
    blk=m.group(0)
    g=lambda k:(re.search(r'\.'+k+r':\s+(\d+)',blk) or [0,'?'])[1]
    print(f"{m.group(1)[:60]:60s} vgpr={g('vgpr_count'):>4} sgpr={g('sgpr_count'):>4} lds={g('group_segment_fixed_size'):>6} scratch={g('private_segment_fixed_size'):>5}")
# per-function fma count
for fm in re.finditer(r'^(_ZN3pcd\w+):\n(.*?)\n\s+s_endpgm', s, re.S|re.M):
    body=fm.group(2)
    n=len(re.findall(r'\bv_(fma|mad|fmac|pk_fma)\w*_f32', body))
    if n: print("FMA-f32 in", fm.group(1)[:60], n)
PY
cp k.s /tmp/last_kernel.s; rm -rf $out
