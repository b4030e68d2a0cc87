"""Randomised parity sweep of the NN paths against the GPU brute force (itself oracle-checked in tests/):
python tools/nn_fuzz.py [seconds] [seed].  Clouds: uniform, planes, integer lattices (exact ties), duplicates,
collinear / coplanar sets, far-from-origin offsets, tiny clouds; queries: near-surface, uniform, on lattice
midpoints, outside the box, NaN/inf rows; batch sizes on both sides of the small-batch threshold; random cell
sizes; grid path, one-launch path, gate-bounded association against the unbounded one."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "colmap-pcd_amd"))
import numpy as np
import pcdhip
from pcdhip import synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
rng = np.random.default_rng(seed0)
ncase = nq = 0


def make_cloud(kind, n):
    if kind == "uniform":
        box = rng.uniform(1, 40, 3)
        return synth.cloud_uniform(n, seed=int(rng.integers(1 << 30)), box=box)[0]
    if kind == "planes":
        return synth.cloud_planes(n, seed=int(rng.integers(1 << 30)), patches=int(rng.integers(3, 30)))[0]
    if kind == "lattice":   # exact ties everywhere
        m = max(2, int(round(n ** (1 / 3))))
        g = np.stack(np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij"), -1).reshape(-1, 3)
        return (g * rng.choice([0.25, 0.5, 1.0])).astype(np.float32)[rng.permutation(g.shape[0])]
    if kind == "duplicates":
        x = synth.cloud_uniform(n, seed=int(rng.integers(1 << 30)), box=np.array([10.0, 10.0, 10.0]))[0]
        x[n // 2:] = x[: n - n // 2]
        return x[rng.permutation(n)]
    if kind == "line":
        t = rng.random(n).astype(np.float32) * 50
        return np.stack([t, 0 * t + 1.5, 0 * t - 2.0], 1).astype(np.float32)
    if kind == "plane_axis":
        x = rng.random((n, 3)).astype(np.float32) * np.array([30, 30, 0], np.float32)
        return x
    raise ValueError(kind)


while time.time() < t_end:
    kind = rng.choice(["uniform", "planes", "lattice", "duplicates", "line", "plane_axis"])
    n = int(rng.choice([50, 3000, 40000, 300000, 2000000], p=[0.2, 0.25, 0.25, 0.25, 0.05]))
    xyz = make_cloud(kind, n)
    if rng.random() < 0.3:
        xyz = (xyz + rng.choice([1e3, -5e3, 1e4]) * rng.random(3)).astype(np.float32)
    n = xyz.shape[0]
    nrm = np.zeros_like(xyz); nrm[:, 2] = 1
    cell = float(rng.choice([0.0, 0.0, 0.1, 0.37, 1.3]))
    c = pcdhip.Cloud(xyz, nrm, raw_lidar_frame=False, cell_size=cell)
    Q = int(rng.choice([1, 7, 500, 20000, 70000, 150000, 300000, 700000], p=[0.1, 0.1, 0.15, 0.2, 0.15, 0.15, 0.1, 0.05]))
    # (>= 262144 queries: rocPRIM's Onesweep in the bookkeeping, below: its merge sort)
    lo, hi = xyz.min(0).astype(np.float64), xyz.max(0).astype(np.float64)
    mode = rng.choice(["near", "uniform", "mid", "outside"])
    if mode == "near":
        q = xyz[rng.integers(0, n, Q)].astype(np.float64) + rng.normal(0, rng.choice([1e-3, 0.05, 0.5]), (Q, 3))
    elif mode == "uniform":
        q = lo + rng.random((Q, 3)) * (hi - lo + 1e-3)
    elif mode == "mid":     # midpoints between cloud points: many exact ties on lattices
        q = 0.5 * (xyz[rng.integers(0, n, Q)].astype(np.float64) + xyz[rng.integers(0, n, Q)].astype(np.float64))
    else:
        q = lo - 5 + rng.random((Q, 3)) * (hi - lo + 10)
    if Q > 10:
        q[rng.integers(0, Q, 3)] = np.nan
        q[rng.integers(0, Q)] = np.inf
    ref = c.nn(q[: min(Q, 20000)], pcdhip.NN_BRUTEFORCE)
    for algo in (pcdhip.NN_AUTO, pcdhip.NN_GRID, pcdhip.NN_FALLBACK_ONLY):
        got = c.nn(q, algo)
        m = min(Q, 20000)
        ok = (np.array_equal(got[0][:m], ref[0]) and np.array_equal(got[1][:m].view(np.uint32), ref[1].view(np.uint32))
              and np.array_equal(got[2][:m], ref[2]))
        if not ok:
            bad = np.nonzero((got[0][:m] != ref[0]) | (got[2][:m] != ref[2]))[0]
            print("MISMATCH", kind, n, cell, Q, mode, "algo", algo, "first", bad[:5], got[0][bad[:5]], ref[0][bad[:5]], flush=True)
            sys.exit(1)
    # gate-bounded association == unbounded association on the accepted rows
    mr = np.round(rng.uniform(0.1, 2.0, Q), 2)
    a0 = c.associate(q, mr, pcdhip.GATE_MAPPER_LOCAL)
    a1 = c.associate(q, mr, pcdhip.GATE_MAPPER_LOCAL | pcdhip.GATE_BOUNDED_SEARCH)
    acc = a0["type"] != 0
    if not (np.array_equal(a0["type"], a1["type"]) and np.array_equal(a0["lidar_xyz"][acc], a1["lidar_xyz"][acc])
            and np.array_equal(a0["dist"][acc], a1["dist"][acc])):
        print("BOUNDED MISMATCH", kind, n, cell, Q, mode, flush=True)
        sys.exit(1)
    c.close()
    ncase += 1; nq += Q
    if ncase % 10 == 0:
        print("cases %d, queries %d, %.0f s left" % (ncase, nq, t_end - time.time()), flush=True)
print("OK: %d cases, %d queries, no mismatch" % (ncase, nq))
