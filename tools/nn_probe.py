"""Quick perf probe of the NN path: python tools/nn_probe.py [N] [Q] [B] [R] [cell]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colmap-pcd_amd"))
import numpy as np
import torch
import pcdhip
from pcdhip import synth

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
Q = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
R = int(sys.argv[4]) if len(sys.argv) > 4 else 2
cell = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
outl = float(sys.argv[6]) if len(sys.argv) > 6 else 0.05
kern = int(os.environ.get("PCD_PROBE_KERNEL", "0"))   # pcd_nn_set_search: 0 clipped brick kernel, 1 clip off, 2 round 3's kernel
pcdhip.set_nn_search(kern)
print("first-stage kernel", kern, flush=True)
t = time.time(); xyz, nrm = synth.cloud_planes(N); q = synth.queries(xyz, Q, outlier_frac=outl, sigma=float(os.environ.get("PCD_PROBE_SIGMA", "0.25"))); print("gen %.1fs" % (time.time() - t), flush=True)
t = time.time(); c = pcdhip.Cloud(xyz, nrm, raw_lidar_frame=False, cell_size=cell); print("build %.2fs" % (time.time() - t), c.info(), flush=True)
dq = torch.from_numpy(q).cuda(); keys = torch.empty(Q, dtype=torch.int64, device="cuda")
pcdhip.set_nn_tuning(B, R, 1)
c.nn_device(dq, Q, keys); torch.cuda.synchronize()
st = c.last_stats(); print("stats", st, "staged/query %.0f pairs/query %.0f" % (st["staged_points"] * 8 / max(Q,1), st["pair_evals"] / Q), flush=True)
pcdhip.set_nn_tuning(B, R, 0)
for _ in range(3): c.nn_device(dq, Q, keys)
torch.cuda.synchronize()
pcdhip.profile_enable(True); pcdhip.profile_reset()
t = time.time()
for _ in range(10): c.nn_device(dq, Q, keys)
torch.cuda.synchronize(); wall = (time.time() - t) / 10
prof = pcdhip.profile_get(); pcdhip.profile_enable(False)
for k, (n, ms) in prof.items(): print("  %-24s %8.3f ms/launch" % (k, ms / n))
tot = sum(ms / n for n, ms in prof.values())
print("kernel sum %.3f ms  wall %.3f ms  -> %.1f M queries/s" % (tot, wall * 1e3, Q / wall / 1e6), flush=True)
# brute-force cross-check on a sample
kb = torch.empty(4096, dtype=torch.int64, device="cuda")
c.nn_device(dq[:4096].contiguous(), 4096, kb, pcdhip.NN_BRUTEFORCE); torch.cuda.synchronize()
print("sample == bruteforce:", bool((kb == keys[:4096]).all()))
