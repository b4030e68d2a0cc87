"""host-side cost of building the BA problem on the device (pcd_ba_create) for the bench scene:
python tools/ba_create_probe.py [cams] [points]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "colmap-pcd_amd"))
import numpy as np, torch, pcdhip
from pcdhip import synth
cams = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
pts = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
scene = synth.ba_scene(cams, pts, seed=11, order="image")
torch.zeros(1, device="cuda")
for rep in range(3):
    t0 = time.perf_counter(); ba = pcdhip.BA(**scene); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("pcd_ba_create: %.1f ms (I=%d P=%d O=%d L=%d)" % ((t1 - t0) * 1e3, ba.I, ba.P, ba.O, ba.L), flush=True)
    ba.close()
