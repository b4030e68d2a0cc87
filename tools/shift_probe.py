import os, sys, time
sys.path.insert(0, "colmap-pcd_amd")
import numpy as np, torch, pcdhip
from pcdhip import synth
N, Q = 10_000_000, 1_000_000
xyz, nrm = synth.cloud_planes(N); q = synth.queries(xyz, Q, seed=99)
dq = torch.from_numpy(q).cuda(); keys = torch.empty(Q, dtype=torch.int64, device="cuda")
ref = None
for cell in (0.0,):
    c = pcdhip.Cloud(xyz, nrm, raw_lidar_frame=False, cell_size=cell)
    print("cell", c.info()["cell_size"], flush=True)
    for (kern, B, R, S) in [(0, 2, 2, 0), (0, 2, 2, 0)]:
        pcdhip.set_nn_search(kern)
        pcdhip.set_brick_shift(S)
        pcdhip.set_nn_tuning(B, R, 1)
        c.nn_device(dq, Q, keys, pcdhip.NN_GRID); torch.cuda.synchronize()
        st = c.last_stats()
        pcdhip.set_nn_tuning(B, R, 0)
        for _ in range(3): c.nn_device(dq, Q, keys, pcdhip.NN_GRID)
        torch.cuda.synchronize()
        pcdhip.profile_enable(True); pcdhip.profile_reset()
        for _ in range(10): c.nn_device(dq, Q, keys, pcdhip.NN_GRID)
        torch.cuda.synchronize()
        prof = pcdhip.profile_get(); pcdhip.profile_enable(False)
        k = keys.clone()
        if ref is None: ref = k
        per = {n: ms / cnt for n, (cnt, ms) in prof.items()}
        print("  kern,B,R,S", (kern, B, R, S), " ".join("%s %.3f" % (n.replace("nn_", ""), v) for n, v in sorted(per.items())),
              "| items %d staged %.0fM pairs/q %.0f fb_q %d | same %s" % (st["brick_groups"], st["staged_points"] / 1e6, st["pair_evals"] / Q, st["fallback_queries"], bool((k == ref).all())), flush=True)
    c.close()
