"""Times the depth-projection association on the bench cloud (10M points), a batch of images."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colmap-pcd_amd"))
import pcdhip
from pcdhip import synth

n = int(os.environ.get("N", 10_000_000)); ni = int(os.environ.get("IMAGES", 64)); nf = int(os.environ.get("FEATS", 4000))
xyz, nrm = synth.cloud_planes(n)
cloud = pcdhip.Cloud(xyz, nrm, raw_lidar_frame=False)
t0 = time.time(); pj = pcdhip.Projector(cloud, min_lidar_proj_dist=0.5); t1 = time.time()
print(f"build: {1e3*(t1-t0):.1f} ms, submaps {pj.num_submaps}")
images, feat = synth.proj_scene(ni, nf, seed=2)
pj.set_new_images(images, feat)
pcdhip.profile_enable(True); pcdhip.profile_reset()
reps = 5
t0 = time.time()
for _ in range(reps):
    found, idx, dist, l6, cam = pj.set_new_images(images, feat)
dt = (time.time() - t0) / reps
print(f"{ni} images x {nf} feats: {1e3*dt:.2f} ms/call, pairs {pj.last_pairs}, found {found.mean():.3f}")
for name, (launches, ms) in pcdhip.profile_get().items():
    print(f"  {name:24s} {launches:5d} launches  {ms/launches:8.3f} ms avg")
