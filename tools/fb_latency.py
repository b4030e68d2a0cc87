"""Single-query latency of the exact pyramid search (k_nn_fallback, one-launch path): distribution over far outliers
(uniform in the cloud's box) and near-surface queries, and the time of small batches of them.  python tools/fb_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "colmap-pcd_amd"))
import numpy as np, torch, pcdhip
from pcdhip import synth
xyz, nrm = synth.cloud_planes(10_000_000)
c = pcdhip.Cloud(xyz, nrm, raw_lidar_frame=False)
rng = np.random.default_rng(1)
lo, hi = xyz.min(0).astype(np.float64), xyz.max(0).astype(np.float64)
far = lo + rng.random((4096, 3)) * (hi - lo)
near = xyz[rng.integers(0, len(xyz), 4096)].astype(np.float64) + rng.normal(0, 0.25, (4096, 3))
keys = torch.empty(4096, dtype=torch.int64, device="cuda")
def t_batch(q, n, reps=20):
    dq = torch.from_numpy(np.ascontiguousarray(q[:n])).cuda()
    for _ in range(3): c.nn_device(dq, n, keys, pcdhip.NN_FALLBACK_ONLY)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): c.nn_device(dq, n, keys, pcdhip.NN_FALLBACK_ONLY)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e6
for name, q in (("far", far), ("near", near)):
    single = []
    for i in range(150):
        single.append(t_batch(q[i:i + 1], 1, reps=5))
    single = np.array(single)
    print(name, "single-query call us: min %.1f med %.1f p90 %.1f max %.1f" % (single.min(), np.median(single), np.quantile(single, .9), single.max()))
    for n in (64, 512, 4096):
        print("   batch of %d: %.1f us" % (n, t_batch(q, n)))
    pcdhip.set_nn_tuning(0, -1, 1)
    dq = torch.from_numpy(np.ascontiguousarray(q)).cuda()
    c.nn_device(dq, 4096, keys, pcdhip.NN_FALLBACK_ONLY); torch.cuda.synchronize()
    st = c.last_stats(); pcdhip.set_nn_tuning(0, -1, 0)
    print("   points/query %.0f" % (st["fallback_points"] / 4096))
