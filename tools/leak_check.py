"""Create / use / destroy every kind of handle in a loop and watch the device memory: python tools/leak_check.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "colmap-pcd_amd"))
import numpy as np, torch, pcdhip
from pcdhip import synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
torch.zeros(1, device="cuda")
xyz, nrm = synth.cloud_planes(400_000, seed=3)
q = synth.queries(xyz, 120_000, seed=4)
mr = synth.max_range_schedule(120_000)
scene = synth.ba_scene(40, 20_000, seed=5, order="image")
imgs, feat = synth.proj_scene(3, 500, seed=6)
d1 = np.random.default_rng(1).integers(0, 255, (700, 128), dtype=np.uint8)


def used():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20


base = None
for r in range(rounds):
    c = pcdhip.Cloud(xyz, nrm, raw_lidar_frame=False)
    c.nn(q); c.nn(q[:5000]); c.associate(q, mr, 0); c.associate(q, mr, pcdhip.GATE_BOUNDED_SEARCH)
    hq, hmr = c.staging(len(q)); hq[:] = q; hmr[:] = mr
    c.associate_staged(len(q), len(q), 0)
    pj = pcdhip.Projector(c); pj.set_new_images(imgs, feat); pj.close()
    c.close()
    ba = pcdhip.BA(**scene); ba.evaluate(("cost", "H_img", "g_img", "H_pt", "g_pt", "W")); ba.evaluate(("residuals", "jac_q")); ba.close()
    pcdhip.sift_match(d1, d1[::-1].copy()); pcdhip.sift_match_batch([d1, d1[:300]], [(0, 1), (1, 0)])
    u = used()
    if r == 2:
        base = u          # scratch of the process-wide SIFT state and allocator pools have settled
    if r % 5 == 0 or r == rounds - 1:
        print("round %3d: %.1f MiB in use" % (r, u), flush=True)
print("growth after round 2: %.1f MiB" % (used() - base))
assert used() - base < 64, "device memory keeps growing"
print("OK")
