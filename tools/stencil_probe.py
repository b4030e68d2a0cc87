"""NN grid path on workload M: brick kernel vs stencil stage cascades, per cell size.
python tools/stencil_probe.py [N] [Q] [cells...]   -> per-scope kernel ms, statistics, and bit-equality of the keys"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colmap-pcd_amd"))
import numpy as np
import torch
import pcdhip
from pcdhip import synth

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
Q = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
cells = [float(x) for x in sys.argv[3:]] or [0.0, 0.2]
xyz, nrm = synth.cloud_planes(N); q = synth.queries(xyz, Q, seed=99)
dq = torch.from_numpy(q).cuda(); keys = torch.empty(Q, dtype=torch.int64, device="cuda")
ref = None
for cell in cells:
    c = pcdhip.Cloud(xyz, nrm, raw_lidar_frame=False, cell_size=cell)
    print("cell", cell, c.info(), flush=True)
    for cfg in [(0, 0, 0, 0), (1, 1, 2, 0), (1, 1, 2, 3), (1, 1, 3, 0), (1, 2, 3, 0), (1, 1, 0, 0), (1, 2, 0, 0)]:
        pcdhip.set_nn_search(*cfg)
        pcdhip.set_nn_tuning(0, -1, 1)
        c.nn_device(dq, Q, keys, pcdhip.NN_GRID); torch.cuda.synchronize()
        st = c.last_stats()
        pcdhip.set_nn_tuning(0, -1, 0)
        for _ in range(3): c.nn_device(dq, Q, keys, pcdhip.NN_GRID)
        torch.cuda.synchronize()
        pcdhip.profile_enable(True); pcdhip.profile_reset()
        t = time.time()
        for _ in range(10): c.nn_device(dq, Q, keys, pcdhip.NN_GRID)
        torch.cuda.synchronize(); wall = (time.time() - t) / 10
        prof = pcdhip.profile_get(); pcdhip.profile_enable(False)
        k = keys.clone()
        if ref is None: ref = k
        same = bool((k == ref).all())
        per = {n: ms / cnt for n, (cnt, ms) in prof.items()}
        print("  cfg %-12s wall %.3f ms | %s | staged/q %.0f fb_q %d fb_pts/q %.0f | keys==ref %s" % (
            cfg, wall * 1e3, " ".join("%s %.3f" % (n.replace("nn_", ""), v) for n, v in sorted(per.items())),
            st["staged_points"] / Q, st["fallback_queries"], st["fallback_points"] / max(st["fallback_queries"], 1), same), flush=True)
        assert same, "keys differ from the first configuration"
    c.close()
pcdhip.set_nn_search(0)
