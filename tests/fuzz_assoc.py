"""(test infrastructure, run by hand; not collected by pytest)  Randomised parity sweep of the association path:
python tests/fuzz_assoc.py [seconds] [seed].
  * pcd_associate (three gate modes, scalar / per-query ranges incl. NaN, negative, 0, huge) vs the oracle restating
    lidar/ply.cc:90-107, lidar_point.cc and the three call-site gates -- clouds with zero / tiny / axis-aligned /
    NaN normals, duplicate points, offsets far from the origin, raw LiDAR frame;
  * pcd_associate_staged (one stream, and the two-chunk pipeline for >= 200 k queries) vs pcd_associate;
  * the two-phase search over 2..5 spatially compact shards (pcd_nn_refine_device) vs the single cloud."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colmap-pcd_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
import pcdhip
from pcdhip import synth, dist as pd
from oracle import pyoracle as oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget


def fail(*a):
    print("MISMATCH", *a, flush=True)
    sys.exit(1)


ncase = 0
while time.time() < t_end:
    n = int(rng.choice([30, 2000, 60000]))
    if rng.random() < 0.5:
        xyz, nrm = synth.cloud_planes(n, seed=int(rng.integers(1 << 30)), patches=int(rng.integers(2, 20)))
    else:
        xyz, nrm = synth.cloud_uniform(n, seed=int(rng.integers(1 << 30)), box=rng.uniform(2, 30, 3))
    xyz = xyz.copy(); nrm = nrm.copy()
    if rng.random() < 0.3:
        xyz = (xyz + rng.choice([500.0, -3000.0, 9000.0]) * rng.random(3)).astype(np.float32)
    k = max(1, n // 20)
    nrm[rng.integers(0, n, k)] = 0.0                                    # |n| < 1e-6 -> rejected (ply.cc:103)
    nrm[rng.integers(0, n, k)] *= 1e-7
    nrm[rng.integers(0, n, k)] = np.array([0, 1, 0], np.float32)        # ground test ratios x/0
    nrm[rng.integers(0, n, k), 0] = np.nan
    if n > 100:
        xyz[n // 2: n // 2 + 20] = xyz[:20]                             # exact duplicates
    c = pcdhip.Cloud(xyz, nrm, raw_lidar_frame=False)
    Q = int(rng.choice([1, 50, 5000, 70000]))
    if rng.random() < 0.1:
        Q = 210000
    q = xyz[rng.integers(0, n, Q)].astype(np.float64) + rng.normal(0, rng.choice([1e-4, 0.05, 0.6]), (Q, 3))
    if Q > 20:
        q[rng.integers(0, Q, 2)] = np.nan
        q[rng.integers(0, Q, 2)] += 1e5
    mode = int(rng.integers(0, 3))
    kind = rng.choice(["per", "scalar"])
    mr = np.round(rng.uniform(0.05, 2.0, Q), 2)
    if Q > 20:
        mr[rng.integers(0, Q, 3)] = [np.nan, -1.0, 0.0]
        mr[rng.integers(0, Q)] = 1e30
    mrv = mr if kind == "per" else float(mr[0])
    # ---- device path vs oracle ----
    idx, sq, found = c.nn(q)
    out6, ok = oracle.search_nearest_neibor(xyz, nrm, idx, found)
    abcd, typ, dist, ang, d2p = oracle.associate(q, out6, ok, None if mode == 2 else mrv, mode)
    for flag in (0, pcdhip.GATE_BOUNDED_SEARCH):
        a = c.associate(q, None if mode == 2 else mrv, mode | flag)
        if not np.array_equal(a["type"], typ):
            fail("type", dict(n=n, Q=Q, mode=mode, kind=kind, flag=flag), np.nonzero(a["type"] != typ)[0][:5])
        acc = typ != 0
        for name, ref in (("lidar_xyz", out6[:, :3]), ("abcd", abcd), ("dist", dist), ("angle", ang)):
            if not np.allclose(a[name][acc], ref[acc], rtol=1e-12, atol=1e-12, equal_nan=True):
                fail(name, dict(n=n, Q=Q, mode=mode, kind=kind, flag=flag))
    # ---- staged host path vs device path ----
    a = c.associate(q, None if mode == 2 else mrv, mode)
    hq, hmr = c.staging(Q)
    hq[:] = q
    if kind == "per":
        hmr[:] = mr
    else:
        hmr[0] = mr[0]
    hits = c.associate_staged(Q, Q if kind == "per" else 1, mode)
    acc = np.nonzero(a["type"])[0]
    if not (len(hits) == len(acc) and np.array_equal(hits["query"], acc) and np.array_equal(hits["type"], a["type"][acc])
            and np.array_equal(hits["lidar_xyz"], a["lidar_xyz"][acc]) and np.array_equal(hits["abcd"], a["abcd"][acc])):
        fail("staged", dict(n=n, Q=Q, mode=mode, kind=kind), len(hits), len(acc))
    # ---- two-phase sharded search vs the single cloud ----
    if n >= 2000:
        S = int(rng.integers(2, 6))
        order = pd.compact_order(xyz)
        cuts = pd.shard_cuts(n, S)
        sh = [pcdhip.Cloud(xyz[order[cuts[r]:cuts[r + 1]]], nrm[order[cuts[r]:cuts[r + 1]]], raw_lidar_frame=False,
                           index_base=int(cuts[r])) for r in range(S)]
        boxes = np.array([s_.info()["bbox_lo"] + s_.info()["bbox_hi"] for s_ in sh])
        fin = np.isfinite(q).all(1)
        home = pd.home_shards(np.where(fin[:, None], q, 0.0), boxes[:, :3], boxes[:, 3:])
        dq = torch.from_numpy(q).cuda()
        k2 = torch.full((Q,), pcdhip.KEY_NONE, dtype=torch.int64, device="cuda")
        for r in range(S):
            ii = torch.from_numpy(np.nonzero(home == r)[0]).cuda()
            if len(ii):
                kh = torch.empty(len(ii), dtype=torch.int64, device="cuda")
                sh[r].nn_device(dq[ii].contiguous(), len(ii), kh)
                k2[ii] = kh
        parts = []
        for r in range(S):
            kr = k2.clone()
            sh[r].nn_refine_device(dq, Q, kr, torch.from_numpy((home == r).astype(np.uint8)).cuda())
            parts.append(kr)
        torch.cuda.synchronize()
        kk = parts[0]
        for p_ in parts[1:]:
            kk = torch.minimum(kk, p_)
        kk = kk.cpu().numpy().view(np.uint64)
        gd = (kk >> np.uint64(32)).astype(np.uint32)
        gi = (kk & np.uint64(0xFFFFFFFF)).astype(np.int64)
        fnd = kk != np.uint64(pcdhip.KEY_NONE)
        if not np.array_equal(fnd, found.astype(bool)):
            fail("sharded found", dict(n=n, Q=Q, S=S))
        if not np.array_equal(gd[fnd], sq.view(np.uint32)[fnd]):
            fail("sharded distance", dict(n=n, Q=Q, S=S), np.nonzero(gd[fnd] != sq.view(np.uint32)[fnd])[0][:5])
        rows = order[gi[fnd]]
        diff = rows != idx[fnd].astype(np.int64)
        if diff.any() and not np.array_equal(xyz[rows[diff]], xyz[idx[fnd][diff]]):
            # a different row is allowed only where another point sits at exactly the same float distance
            qf = q[fnd][diff].astype(np.float32)
            def fd(p):
                dx, dy, dz = qf[:, 0] - p[:, 0], qf[:, 1] - p[:, 1], qf[:, 2] - p[:, 2]
                return (dx * dx + dy * dy) + dz * dz
            if not np.array_equal(fd(xyz[rows[diff]]).view(np.uint32), fd(xyz[idx[fnd][diff]]).view(np.uint32)):
                fail("sharded index", dict(n=n, Q=Q, S=S))
        for s_ in sh:
            s_.close()
    c.close()
    ncase += 1
    if ncase % 20 == 0:
        print("cases %d, %.0f s left" % (ncase, t_end - time.time()), flush=True)
print("OK: %d cases, no mismatch" % ncase)
