"""What one rank of an N-GPU strong-scaling run does, timed on ONE GPU: the full cloud, queries [0, Q/N), the tracks
p = 0 mod N of the one BA scene (pcdhip/dist.py) -- i.e. bench.py's per-rank step without the collectives.
python tools/scaling_probe.py [N ...]   (predicted speed-up = t(1) / t(N), before all-reduce latency)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colmap-pcd_amd"))
import numpy as np
import torch
import pcdhip
from pcdhip import synth, dist as pd

SPATIAL = os.environ.get("PROBE_SPATIAL", "0") == "1"   # queries put in a spatially compact order before the split
Ns = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
xyz, nrm = synth.cloud_planes(10_000_000)
q_all = synth.queries(xyz, 1_000_000, seed=99)
mr_all = synth.max_range_schedule(1_000_000, seed=5)
if SPATIAL:
    perm = pd.compact_order(q_all)
    q_all, mr_all = q_all[perm], mr_all[perm]
scene = synth.ba_scene(1000, 1_000_000, seed=11, order="image")
cloud = pcdhip.Cloud(xyz, nrm, raw_lidar_frame=False)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
t1 = None
for N in Ns:
    lo, hi = pd.shard_range(1_000_000, 0, N)
    Q = hi - lo
    sub = scene if N == 1 else pd.shard_tracks(scene, 0, N)[0]
    ba = pcdhip.BA(**sub)
    dq = torch.from_numpy(np.ascontiguousarray(q_all[lo:hi])).to(dev)
    dmr = torch.from_numpy(np.ascontiguousarray(mr_all[lo:hi])).to(dev)
    keys = torch.empty(Q, dtype=torch.int64, device=dev)
    f64 = lambda *s: torch.empty(*s, dtype=torch.float64, device=dev)
    aout = dict(lidar_xyz=f64(Q, 3), abcd=f64(Q, 4), type=torch.empty(Q, dtype=torch.uint8, device=dev), dist=f64(Q), angle=f64(Q))
    I, P, O = ba.I, ba.P, ba.O
    blocks = f64(I * 42 + 1)
    full = dict(cost=blocks[I * 42:], H_img=blocks[:I * 36], g_img=blocks[I * 36:I * 42], H_pt=f64(P, 9), g_pt=f64(P, 3), W=f64(O, 18))
    cr = f64(1)
    BOUNDED = os.environ.get("PROBE_BOUNDED", "0") == "1"
    def step():
        if BOUNDED:
            cloud.associate_device(dq, Q, dmr, Q, pcdhip.GATE_MAPPER_LOCAL | pcdhip.GATE_BOUNDED_SEARCH, aout, None, stream)
        else:
            cloud.nn_device(dq, Q, keys, pcdhip.NN_AUTO, stream)
            cloud.associate_device(dq, Q, dmr, Q, pcdhip.GATE_MAPPER_LOCAL, aout, keys, stream)
        ba.evaluate_device(full, stream)
        ba.evaluate_device(dict(cost=cr), stream)
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 20
    pcdhip.profile_enable(True); pcdhip.profile_reset()   # per-kernel breakdown in its own pass (events cost stream time)
    for _ in range(20): step()
    torch.cuda.synchronize()
    prof = pcdhip.profile_get(); pcdhip.profile_enable(False)
    t1 = t1 or t
    print("N=%d: per-rank step %.3f ms (Q=%d, obs=%d) -> speed-up before collectives %.2fx   %s" % (
        N, t * 1e3, Q, O, t1 / t, {k: round(ms / n, 3) for k, (n, ms) in sorted(prof.items())}), flush=True)
    ba.close()
