"""(test infrastructure, run by hand; not collected by pytest)  Randomised parity sweep of the depth-projection association
(lidar/pcd_projection.cc) against the oracle: python tests/fuzz_proj.py [seconds] [seed].  Random clouds, image counts,
feature counts, image sizes, OPENCV parameters, scale / splat / submap options."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colmap-pcd_amd")); sys.path.insert(0, ROOT)
import numpy as np
import pcdhip
from pcdhip import synth
from oracle import pyoracle as oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
ncase = 0
while time.time() < t_end:
    n = int(rng.choice([500, 30000, 200000]))
    xyz, nrm = (synth.cloud_planes(n, seed=int(rng.integers(1 << 30))) if rng.random() < 0.6
                else synth.cloud_uniform(n, seed=int(rng.integers(1 << 30))))
    w, h = [(4032, 3024), (2000, 1500), (640, 480), (1201, 907)][int(rng.integers(0, 4))]
    f = rng.uniform(0.6, 1.4) * w
    prm = [f, f * rng.uniform(0.97, 1.03), w / 2 + rng.normal(0, 20), h / 2 + rng.normal(0, 20),
           rng.normal(0, 0.05), rng.normal(0, 0.02), rng.normal(0, 3e-4), rng.normal(0, 3e-4)]
    ni, nf = int(rng.choice([1, 3, 9])), int(rng.choice([1, 200, 3000]))
    images, feat = synth.proj_scene(ni, nf, seed=int(rng.integers(1 << 30)), width=w, height=h, params=prm)
    okw = dict(depth_image_scale=float(rng.choice([0.1, 0.2, 0.25, 0.5])), max_proj_scale=int(rng.choice([3, 6, 10])),
               min_proj_scale=int(rng.choice([1, 2])), min_proj_dist=float(rng.choice([1.0, 3.0, 8.0])),
               submap=float(rng.choice([1.0, 2.0, 5.0])), choose_meter=float(rng.choice([10.0, 25.0, 60.0])),
               min_lidar_proj_dist=float(rng.choice([0.0, 0.5, 2.0])))
    oo = oracle.proj_options(**okw)
    coeffs = oracle.proj_scale_coeffs(oo, images[0]["params"][0], images[0]["params"][1])
    exp = oracle.proj_images(xyz, nrm, oo, coeffs, images, feat)
    cloud = pcdhip.Cloud(xyz, nrm, raw_lidar_frame=False)
    pj = pcdhip.Projector(cloud, depth_image_scale=oo.depth_image_scale, max_proj_scale=oo.max_proj_scale,
                          min_proj_scale=oo.min_proj_scale, min_proj_dist=oo.min_proj_dist, submap_length=oo.submap_length,
                          submap_width=oo.submap_width, submap_height=oo.submap_height, choose_meter=oo.choose_meter,
                          min_lidar_proj_dist=oo.min_lidar_proj_dist)
    found, index, dist, l6, cam = pj.set_new_images(images, feat)
    ok = (np.array_equal(found, exp[0]) and np.array_equal(index, exp[1]) and np.array_equal(dist.view(np.uint32), exp[2].view(np.uint32))
          and np.array_equal(l6, exp[3]) and np.allclose(cam, exp[4], rtol=1e-12, atol=0, equal_nan=True) and pj.last_pairs == exp[5])
    if not ok:
        print("MISMATCH", dict(n=n, w=w, h=h, ni=ni, nf=nf, **okw), int((found != exp[0]).sum()), int((index != exp[1]).sum()), flush=True)
        sys.exit(1)
    pj.close(); cloud.close()
    ncase += 1
    if ncase % 20 == 0:
        print("cases %d, %.0f s left" % (ncase, t_end - time.time()), flush=True)
print("OK: %d cases, no mismatch" % ncase)
