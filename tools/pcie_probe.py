"""PCIe-inclusive rate of the host-buffer entry points (pageable numpy buffers -> pcd_associate -> numpy):
python tools/pcie_probe.py [cloud_points] [queries]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colmap-pcd_amd"))
import numpy as np
import pcdhip
from pcdhip import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
q = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
xyz, nrm = synth.cloud_planes(n)
raw = synth.visual_to_raw(xyz, nrm)
t0 = time.time(); cloud = pcdhip.Cloud(*raw); t1 = time.time()
print(f"cloud create from host arrays (upload + transform + index): {1e3*(t1-t0):.1f} ms")
Q = synth.queries(xyz, q)
mr = synth.max_range_schedule(q)
cloud.associate(Q, mr)
reps = 5
t0 = time.time()
for _ in range(reps):
    cloud.associate(Q, mr)
dt = (time.time() - t0) / reps
print(f"pcd_associate host->host, {q} queries: {1e3*dt:.2f} ms  = {q/dt/1e6:.1f} M queries/s (PCIe-inclusive)")
t0 = time.time()
for _ in range(reps):
    cloud.nn(Q)
dt = (time.time() - t0) / reps
print(f"pcd_nn_query host->host, {q} queries: {1e3*dt:.2f} ms  = {q/dt/1e6:.1f} M queries/s (PCIe-inclusive)")
