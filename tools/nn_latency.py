"""Per-call latency of the NN + association path for small batches (config A of BASELINE.md: 2 M-point cloud,
20 k queries -- the incremental mapper's local-BA calls, sfm/incremental_mapper.cc:1155-1165), AUTO (grid path)
against FALLBACK_ONLY (one launch): python tools/nn_latency.py [N] [Q ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colmap-pcd_amd"))
import numpy as np
import torch
import pcdhip
from pcdhip import synth

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
Qs = [int(float(a)) for a in sys.argv[2:]] or [1000, 5000, 20000, 100000]
xyz, nrm = synth.cloud_planes(N)
c = pcdhip.Cloud(xyz, nrm, raw_lidar_frame=False)
print("cloud", N, c.info(), flush=True)
for Q in Qs:
    q = synth.queries(xyz, Q)
    dq = torch.from_numpy(q).cuda(); keys = torch.empty(Q, dtype=torch.int64, device="cuda")
    for name, algo in (("AUTO", pcdhip.NN_AUTO), ("GRID", pcdhip.NN_GRID), ("FALLBACK_ONLY", pcdhip.NN_FALLBACK_ONLY)):
        try:
            for _ in range(5): c.nn_device(dq, Q, keys, algo)
        except pcdhip.PcdError:
            continue
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(50): c.nn_device(dq, Q, keys, algo)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 50
        print("Q=%7d %-14s %8.1f us per call  (%.1f M queries/s)" % (Q, name, dt * 1e6, Q / dt / 1e6), flush=True)
