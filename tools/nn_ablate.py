"""timing-only ablations of k_nn_brick_clip (results are wrong while a flag is set; PCD_BRICK_BLOCKS=1..4 in the
environment = workgroups per CU, i.e. wavefronts per SIMD, of the brick kernel).  Needs a variant of the library
built with -DPCD_ABLATE (the shipped one has no ablation branches):
  PCDHIP_LIB=colmap-pcd_amd/variants/libpcdhip_ablate.so python tools/nn_ablate.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colmap-pcd_amd"))
import torch  # noqa: E402
import pcdhip  # noqa: E402
from pcdhip import synth  # noqa: E402

N, Q = 10_000_000, 1_000_000
xyz, nrm = synth.cloud_planes(N)
q = synth.queries(xyz, Q)
c = pcdhip.Cloud(xyz, nrm, raw_lidar_frame=False)
dq = torch.from_numpy(q).cuda()
keys = torch.empty(Q, dtype=torch.int64, device="cuda")
F = 0x800  # never hand anything to the fallback (keeps the brick kernel's control flow comparable)
# k_nn_brick_clip (csrc/brick_clip_kernel.h): 0x100 no compare, 0x400 no reductions, 0x1000 no stage B, 0x10000 no clip
# arithmetic (whole region), 0x20000 no region bound, 0x40000 no stage A
for name, fl in [("full", 0), ("full, no fallback list", F), ("no compare", F | 0x100), ("no reductions", F | 0x400),
                 ("no stage B", F | 0x1000), ("no stage A", F | 0x40000), ("no region bound", F | 0x20000),
                 ("no clip arithmetic (whole region)", F | 0x10000), ("no stage B, no stage A", F | 0x41000),
                 ("no stage B / A / reductions / bound", F | 0x61400),
                 ("no compare, no reductions, no bound", F | 0x20500),
                 ("no DMA (stale LDS)", F | 0x4000), ("no LDS read", F | 0x8000), ("no DMA, no LDS read", F | 0xC000),
                 ("no range select", F | 0x80000), ("no DMA, no LDS read, no select", F | 0x8C000),
                 ("no DMA, no LDS read, no select, no compare", F | 0x8C100), ("full", 0)]:
    pcdhip.set_nn_tuning(0, -1, fl)
    for _ in range(3):
        c.nn_device(dq, Q, keys)
    torch.cuda.synchronize()
    pcdhip.profile_enable(True)
    pcdhip.profile_reset()
    for _ in range(10):
        c.nn_device(dq, Q, keys)
    torch.cuda.synchronize()
    p = pcdhip.profile_get()
    pcdhip.profile_enable(False)
    print("%-30s nn_brick %.3f ms   fallback %.3f ms" % (name, p["nn_brick"][1] / p["nn_brick"][0],
                                                          p["nn_fallback"][1] / p["nn_fallback"][0]), flush=True)
