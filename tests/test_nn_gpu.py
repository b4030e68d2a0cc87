"""GPU parity of the nearest-neighbour path (through the C ABI) against the CPU oracle.

Bar: bit-exact -- same index, same float32 squared-distance bits, same found flag
(reference semantics: lidar/kdtree.cc:10-21 + lidar/ply.cc:90-93; ties -> lowest index).
"""
import numpy as np
import pytest

from pcdhip import synth

pytestmark = pytest.mark.gpu

ALGOS = ["AUTO", "GRID", "BRUTEFORCE", "FALLBACK_ONLY"]   # AUTO = FALLBACK_ONLY at these batch sizes


def _algo(pcdhip, name):
    return getattr(pcdhip, "NN_" + name)


def _check_exact(got, exp, what=""):
    gi, gd, gf = got
    ei, ed, ef = exp
    assert np.array_equal(gf, ef), f"{what}: found flags differ at {np.nonzero(gf != ef)[0][:10]}"
    bad = np.nonzero((gi != ei) | (gd.view(np.uint32) != ed.view(np.uint32)))[0]
    assert bad.size == 0, (f"{what}: {bad.size} mismatches, first {bad[:5]}: idx {gi[bad[:5]]} vs {ei[bad[:5]]}, "
                           f"d {gd[bad[:5]]} vs {ed[bad[:5]]}")


def _clouds():
    rng = np.random.default_rng(3)
    out = {}
    out["uniform"] = synth.cloud_uniform(40000, seed=7, box=np.array([20.0, 5.0, 20.0]))
    out["planes"] = synth.cloud_planes(50000, seed=20240601, patches=24)
    # duplicates (exact ties) + clustered points
    x, n = synth.cloud_uniform(20000, seed=8, box=np.array([10.0, 10.0, 10.0]))
    x[5000:10000] = x[0:5000]
    x[15000:] = x[15000] + rng.normal(0, 1e-3, (5000, 3)).astype(np.float32)
    out["duplicates"] = (x, n)
    return out


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("name", ["uniform", "planes", "duplicates"])
def test_nn_parity_small(gpu, oracle, name, algo):
    xyz, nrm = _clouds()[name]
    q = synth.queries(xyz, 4000, seed=99)
    # some queries exactly on cloud points (ties among duplicates -> lowest index must win)
    q[:200] = xyz[np.random.default_rng(1).integers(0, xyz.shape[0], 200)].astype(np.float64)
    c = gpu.Cloud(xyz, nrm, raw_lidar_frame=False)
    got = c.nn(q, _algo(gpu, algo))
    exp = oracle.nn_bruteforce(xyz, q)
    _check_exact(got, exp, f"{name}/{algo}")
    c.close()


@pytest.mark.parametrize("cell", [0.05, 0.3, 2.0, 50.0])
def test_nn_parity_any_cell_size(gpu, oracle, cell):
    """exactness must not depend on the grid resolution (pruning is conservative)"""
    xyz, nrm = synth.cloud_planes(30000, seed=5, patches=12)
    q = synth.queries(xyz, 3000, seed=17, sigma=0.6)
    c = gpu.Cloud(xyz, nrm, raw_lidar_frame=False, cell_size=cell)
    exp = oracle.nn_bruteforce(xyz, q)
    for algo in ("GRID", "FALLBACK_ONLY"):
        _check_exact(c.nn(q, _algo(gpu, algo)), exp, f"cell={cell}/{algo}")
    c.close()


def test_raw_frame_transform_and_nan_rows(gpu, oracle):
    """lidar/ply.cc:33-57: axis swap, rows with any NaN dropped, order kept; indices are post-filter."""
    xyz_v, nrm_v = synth.cloud_uniform(5000, seed=21, box=np.array([8.0, 3.0, 8.0]))
    raw_xyz, raw_nrm = synth.visual_to_raw(xyz_v, nrm_v)
    rng = np.random.default_rng(4)
    raw_xyz[rng.integers(0, 5000, 60), rng.integers(0, 3, 60)] = np.nan
    raw_nrm[rng.integers(0, 5000, 40), rng.integers(0, 3, 40)] = np.nan
    exp_xyz, exp_nrm = oracle.direction_trans(raw_xyz, raw_nrm)
    c = gpu.Cloud(raw_xyz, raw_nrm, raw_lidar_frame=True)
    assert len(c) == exp_xyz.shape[0] < 5000
    got_xyz, got_nrm = c.download()
    assert np.array_equal(got_xyz.view(np.uint32), exp_xyz.view(np.uint32))
    assert np.array_equal(got_nrm.view(np.uint32), exp_nrm.view(np.uint32))
    q = synth.queries(exp_xyz, 1500, seed=2)
    _check_exact(c.nn(q), oracle.nn_bruteforce(exp_xyz, q), "raw frame")
    # AoS32 layout of lidarpt::Point (lidar/pt_type.h:14-30)
    aos = np.zeros((5000, 8), np.float32)
    aos[:, 0:3] = raw_xyz
    aos[:, 4:7] = raw_nrm
    c2 = gpu.Cloud(aos, layout=gpu.LAYOUT_AOS32, raw_lidar_frame=True)
    assert len(c2) == len(c)
    _check_exact(c2.nn(q), oracle.nn_bruteforce(exp_xyz, q), "aos32")
    c.close(); c2.close()


def test_edge_cases(gpu, oracle):
    z3 = np.zeros((0, 3), np.float32)
    q = np.array([[0.0, 0, 0], [1e30, 0, 0], [np.nan, 0, 0], [np.inf, 1, 1], [1e-40, -1e-40, 0]])
    # empty cloud: Kdtree::GetClosestPoint returns false
    c = gpu.Cloud(z3, z3, raw_lidar_frame=False)
    for a in ALGOS:
        i, d, f = c.nn(q, _algo(gpu, a))
        assert not f.any() and (i == 0xFFFFFFFF).all()
    assert c.nn(np.zeros((0, 3)))[0].shape == (0,)
    c.close()
    # all rows NaN -> empty after the filter
    allnan = np.full((7, 3), np.nan, np.float32)
    c = gpu.Cloud(allnan, allnan, raw_lidar_frame=True)
    assert len(c) == 0 and not c.nn(q)[2].any()
    c.close()
    # single point, Inf rows keep their index but never win, far / non-finite queries
    xyz = np.array([[np.inf, 0, 0], [1.0, 2.0, 3.0], [0, -np.inf, 0], [1.0, 2.0, 3.0]], np.float32)
    c = gpu.Cloud(xyz, np.ones_like(xyz), raw_lidar_frame=False)
    exp = oracle.nn_bruteforce(xyz, q)
    for a in ALGOS:
        _check_exact(c.nn(q, _algo(gpu, a)), exp, f"edge/{a}")
    i, d, f = c.nn(q)
    assert list(f) == [1, 0, 0, 0, 1] and i[0] == 1 and i[4] == 1   # 1e30: squared distance overflows -> not found
    c.close()
    # degenerate extents: all points on a line / in one cell
    line = np.zeros((3000, 3), np.float32)
    line[:, 0] = np.linspace(0, 30, 3000)
    c = gpu.Cloud(line, np.ones_like(line), raw_lidar_frame=False)
    qq = synth.queries(line, 500, seed=3, sigma=2.0)
    for a in ALGOS:
        _check_exact(c.nn(qq, _algo(gpu, a)), oracle.nn_bruteforce(line, qq), f"line/{a}")
    c.close()


def test_queries_outside_bbox_and_far(gpu, oracle):
    xyz, nrm = synth.cloud_planes(20000, seed=9, patches=6)
    rng = np.random.default_rng(12)
    q = (rng.random((2000, 3)) - 0.5) * 600.0      # mostly far outside the cloud's bounding box
    c = gpu.Cloud(xyz, nrm, raw_lidar_frame=False)
    exp = oracle.nn_bruteforce(xyz, q)
    for a in ALGOS:
        _check_exact(c.nn(q, _algo(gpu, a)), exp, f"far/{a}")
    c.close()


def test_golden_fixture(gpu):
    """committed vectors (tests/golden/make_golden.py, produced with the CPU oracle)"""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "nn_small.npz"))
    c = gpu.Cloud(g["xyz"], g["nrm"], raw_lidar_frame=False)
    for a in ALGOS:
        i, d, f = c.nn(g["q"], _algo(gpu, a))
        assert np.array_equal(i, g["idx"]) and np.array_equal(d.view(np.uint32), g["sqdist_bits"])
        assert np.array_equal(f, g["found"])
    c.close()


def test_midsize_vs_kdtree_oracle_and_gpu_bruteforce(gpu, oracle):
    """2 M points / 200 k queries: grid kernels == GPU brute force everywhere, == exact CPU KD-tree on a sample,
    plus size-independent properties (returned distance is the distance to the returned index; no staged or
    sampled point is closer)."""
    xyz, nrm = synth.cloud_planes(2_000_000, seed=20240601)
    q = synth.queries(xyz, 200_000, seed=99)
    c = gpu.Cloud(xyz, nrm, raw_lidar_frame=False)
    gi, gd, gf = c.nn(q)
    bi, bd, bf = c.nn(q[:20000], gpu.NN_BRUTEFORCE)
    _check_exact((gi[:20000], gd[:20000], gf[:20000]), (bi, bd, bf), "grid vs gpu brute force")
    kd = oracle.KDTree(xyz)
    sel = np.random.default_rng(0).choice(q.shape[0], 20000, replace=False)
    _check_exact((gi[sel], gd[sel], gf[sel]), kd.query(q[sel]), "grid vs cpu kd-tree")
    # property: sqdist is exactly the float distance to the returned point
    qf = q.astype(np.float32)
    p = xyz[gi]
    dx, dy, dz = qf[:, 0] - p[:, 0], qf[:, 1] - p[:, 1], qf[:, 2] - p[:, 2]
    d = (dx * dx + dy * dy) + dz * dz
    assert np.array_equal(d.view(np.uint32), gd.view(np.uint32))
    st = c.last_stats()
    c.close()


def test_every_query_unproven_inside_the_grid(gpu, oracle):
    """two slabs 40 m apart and 300 k queries in the empty space between them: inside the grid, nothing within any
    brick's halo, so EVERY query goes through the brick kernel's chunked fallback list (chunks left with unused
    slots, list capacity, squeeze) and is answered by the exact fallback.  Bit-exact against the CPU KD-tree."""
    rng = np.random.default_rng(17)
    a = rng.random((150_000, 3)).astype(np.float32) * np.array([0.5, 30.0, 30.0], np.float32)
    b = a.copy(); b[:, 0] += 40.0
    xyz = np.concatenate([a, b]); nrm = np.zeros_like(xyz); nrm[:, 0] = 1.0
    q = rng.random((300_000, 3)) * np.array([30.0, 30.0, 30.0]) + np.array([5.0, 0.0, 0.0])
    c = gpu.Cloud(xyz, nrm, raw_lidar_frame=False)
    got = c.nn(q, gpu.NN_GRID)
    st = c.last_stats()
    kd = oracle.KDTree(xyz)
    sel = rng.choice(q.shape[0], 30000, replace=False)
    gi, gd, gf = got
    _check_exact((gi[sel], gd[sel], gf[sel]), kd.query(q[sel]), "all-unproven grid path vs cpu kd-tree")
    assert gf.all()
    c.close()


@pytest.mark.parametrize("N,Q", [(10_000_000, 1_000_000), (20_000_000, 2_000_000)],
                         ids=["config M: 10M cloud / 1M queries", "config C: 20M cloud / 2M queries"])
def test_full_size_properties(gpu, N, Q):
    """BASELINE.json's sizes (M: 10 M-point cloud / 1 M queries; C: 20 M / 2 M): too large for the CPU oracle in a test, so the
    result is checked through properties that do not depend on the size:
      * the returned squared distance is bit-for-bit the FLANN float distance to the returned row;
      * a 4096-query sample equals the GPU brute force over all 10 M rows (itself oracle-checked above);
      * querying cloud rows themselves gives distance 0 and an index whose row has identical coordinates,
        never a higher index than the queried row (ties go to the lowest index);
      * splitting the cloud in two interleaved shards and taking the per-query minimum of the packed keys
        reproduces the single-cloud keys (the cloud-sharded multi-GPU path, dist.py)."""
    import torch
    xyz, nrm = synth.cloud_planes(N)
    q = synth.queries(xyz, Q)
    c = gpu.Cloud(xyz, nrm, raw_lidar_frame=False)
    dq = torch.from_numpy(q).cuda()
    keys = torch.empty(Q, dtype=torch.int64, device="cuda")
    c.nn_device(dq, Q, keys)
    torch.cuda.synchronize()
    k = keys.cpu().numpy().view(np.uint64)
    assert (k != np.uint64(gpu.KEY_NONE)).all()
    gi = (k & np.uint64(0xFFFFFFFF)).astype(np.int64)
    gd = (k >> np.uint64(32)).astype(np.uint32)
    qf = q.astype(np.float32)
    p = xyz[gi]
    dx, dy, dz = qf[:, 0] - p[:, 0], qf[:, 1] - p[:, 1], qf[:, 2] - p[:, 2]
    d = (dx * dx + dy * dy) + dz * dz
    assert np.array_equal(d.view(np.uint32), gd)
    # sample vs brute force over the whole cloud
    kb = torch.empty(4096, dtype=torch.int64, device="cuda")
    sel = torch.from_numpy(np.random.default_rng(1).choice(Q, 4096, replace=False)).cuda()
    c.nn_device(dq[sel].contiguous(), 4096, kb, gpu.NN_BRUTEFORCE)
    torch.cuda.synchronize()
    assert torch.equal(kb, keys[sel])
    # cloud rows as queries
    rows = np.random.default_rng(2).choice(N, 200_000, replace=False)
    ri, rd, rf = c.nn(xyz[rows].astype(np.float64))
    assert rf.all() and not rd.any()
    assert np.array_equal(xyz[ri], xyz[rows]) and (ri <= rows).all()
    c.close()
    # two interleaved shards, combined by the minimum of the keys
    Qs = 200_000
    ks = []
    for r in range(2):
        cs = gpu.Cloud(xyz[r::2], nrm[r::2], raw_lidar_frame=False, index_base=r, index_stride=2)
        kk = torch.empty(Qs, dtype=torch.int64, device="cuda")
        cs.nn_device(dq[:Qs].contiguous(), Qs, kk)
        torch.cuda.synchronize()
        ks.append(kk.clone())
        cs.close()
    assert torch.equal(torch.minimum(ks[0], ks[1]), keys[:Qs])
    # two spatially compact shards, two-phase search (pcd_nn_refine_device): same keys up to the row permutation
    from pcdhip import dist as pd
    order = pd.compact_order(xyz)
    inv = np.empty(N, np.int64); inv[order] = np.arange(N)
    cuts = pd.shard_cuts(N, 2)
    sh = [gpu.Cloud(xyz[order[cuts[r]:cuts[r + 1]]], nrm[order[cuts[r]:cuts[r + 1]]], raw_lidar_frame=False,
                    index_base=cuts[r]) for r in range(2)]
    boxes = np.array([s_.info()["bbox_lo"] + s_.info()["bbox_hi"] for s_ in sh])
    home = pd.home_shards(q[:Qs], boxes[:, :3], boxes[:, 3:])
    k2 = torch.full((Qs,), gpu.KEY_NONE, dtype=torch.int64, device="cuda")
    for r in range(2):                                   # phase 1: home shard only
        idx = torch.from_numpy(np.nonzero(home == r)[0]).cuda()
        kh = torch.empty(len(idx), dtype=torch.int64, device="cuda")
        sh[r].nn_device(dq[:Qs][idx].contiguous(), len(idx), kh)
        k2[idx] = kh
    parts = []
    for r in range(2):                                   # phase 2: the foreign queries a shard cannot rule out
        kr = k2.clone()
        sh[r].nn_refine_device(dq[:Qs].contiguous(), Qs, kr, torch.from_numpy((home == r).astype(np.uint8)).cuda())
        parts.append(kr)
    torch.cuda.synchronize()
    kk = torch.minimum(parts[0], parts[1]).cpu().numpy().view(np.uint64)
    ref = keys[:Qs].cpu().numpy().view(np.uint64)
    assert np.array_equal(kk >> np.uint64(32), ref >> np.uint64(32))                 # same distances
    same_pt = order[(kk & np.uint64(0xFFFFFFFF)).astype(np.int64)]                   # back to original rows
    ref_i = (ref & np.uint64(0xFFFFFFFF)).astype(np.int64)
    diff = same_pt != ref_i
    # a different row only on an exact tie of the FLOAT distance to the query (the compact order has its own lowest
    # index): the distance bits are equal for every row (checked above), so what remains to exclude is a wrong row that
    # happens to carry the same bits -- recompute both rows' distances to the query in FLANN's arithmetic, zero tolerance
    def l2(qf, p):
        d = (qf[:, 0] - p[:, 0]) * (qf[:, 0] - p[:, 0])
        d = d + (qf[:, 1] - p[:, 1]) * (qf[:, 1] - p[:, 1])
        return d + (qf[:, 2] - p[:, 2]) * (qf[:, 2] - p[:, 2])
    qf = q[:Qs].astype(np.float32)[diff]
    da, db = l2(qf, xyz[same_pt[diff]]), l2(qf, xyz[ref_i[diff]])
    assert np.array_equal(da.view(np.uint32), db.view(np.uint32)), "a differing row that is not an exact distance tie"
    assert np.array_equal(da.view(np.uint32), (ref[diff] >> np.uint64(32)).astype(np.uint32))
    for s_ in sh:
        s_.close()


@pytest.mark.parametrize("offset", [1e3, 1e5, 3e6])
def test_far_from_origin(gpu, oracle, offset):
    """coordinates with few mantissa bits left for the cell arithmetic (float spacing 0.25 m at 3e6): the
    binning slack sends more queries to the exact fallback, the result stays bit-exact"""
    xyz, nrm = synth.cloud_planes(300_000, seed=3)
    xyz = (xyz.astype(np.float64) + offset).astype(np.float32)
    q = synth.queries(xyz, 30_000, seed=4)
    c = gpu.Cloud(xyz, nrm, raw_lidar_frame=False)
    bf = c.nn(q, gpu.NN_BRUTEFORCE)
    _check_exact(c.nn(q, gpu.NN_GRID), bf, f"grid vs brute force at offset {offset}")
    _check_exact(c.nn(q), bf, f"auto vs brute force at offset {offset}")
    # ... and the GPU brute force itself against the oracle on a sample (not only GPU against GPU)
    sel = np.random.default_rng(6).choice(len(q), 600, replace=False)
    exp = oracle.nn_bruteforce(xyz, q[sel])
    _check_exact(tuple(a[sel] for a in bf), exp, f"brute force vs oracle at offset {offset}")
    c.close()


def test_config_A_one_launch_path(gpu, oracle):
    """BASELINE config A (Smith Hall 25-like): 2 M-point cloud, 20 k queries per call -- PCD_NN_AUTO takes the one-launch
    path (k_nn_fallback<1 / 2>) at this batch size.  Plain search against the KD-tree oracle (bit-exact), gate-bounded
    association against the unbounded one + gate."""
    xyz, nrm = synth.cloud_planes(2_000_000)
    q = synth.queries(xyz, 20_000, seed=3)
    c = gpu.Cloud(xyz, nrm, raw_lidar_frame=False)
    kd = oracle.KDTree(xyz)
    exp = kd.query_mt(q, 8)
    _check_exact(c.nn(q, gpu.NN_AUTO), exp, "config A, one launch, plain")
    _check_exact(c.nn(q, gpu.NN_GRID), exp, "config A, grid path")
    mr = synth.max_range_schedule(20_000, seed=3)
    a0 = c.associate(q, mr, gpu.GATE_MAPPER_LOCAL)
    a1 = c.associate(q, mr, gpu.GATE_MAPPER_LOCAL | gpu.GATE_BOUNDED_SEARCH)
    assert np.array_equal(a0["type"], a1["type"])
    acc = a0["type"] != 0
    assert acc.sum() > 10_000
    for k in ("lidar_xyz", "abcd", "dist", "angle", "nn_idx"):
        assert np.array_equal(a0[k][acc], a1[k][acc]), k
    # the accepted associations against the oracle's epilogue on the KD-tree's winners
    out6, ok = oracle.search_nearest_neibor(xyz, nrm, exp[0], exp[2])
    abcd, typ, dist, ang, d2p = oracle.associate(q, out6, ok, mr, 0)
    assert np.array_equal(a0["type"], typ)
    np.testing.assert_allclose(a0["abcd"][acc], abcd[acc], rtol=1e-12, atol=1e-300)
    c.close()


def _hard_cases():
    rng = np.random.default_rng(31)
    cases = dict(_clouds())
    dense, dn = synth.cloud_uniform(30000, seed=5, box=np.array([6.0, 6.0, 6.0]))
    dense[:12000] = dense[0] + rng.normal(0, 2e-3, (12000, 3)).astype(np.float32)   # 12 k points inside one cell
    cases["dense_cell"] = (dense, dn)
    lat = np.stack(np.meshgrid(np.arange(40), np.arange(30), np.arange(20), indexing="ij"), -1).reshape(-1, 3).astype(np.float32) * 0.25
    cases["lattice"] = (lat, np.ones_like(lat))
    # a thin sheet: every brick holds a few points, the balls of most queries reach into the halo rows
    sheet = (rng.random((40000, 3)) * np.array([30.0, 30.0, 0.02])).astype(np.float32)
    cases["sheet"] = (sheet, np.tile(np.array([0, 0, 1], np.float32), (40000, 1)))
    return cases


@pytest.mark.parametrize("kernel", [0, 1, 2])
def test_brick_kernel_variants(gpu, oracle, kernel):
    """First stage of the grid path: 0 = the clipped brick kernel (csrc/brick_clip_kernel.h: the brick's own cells first,
    then only the quad-row parts the queries' balls touch), 1 = the same kernel with the clip switched off, 2 = round
    3's whole-region kernel.  All three must give the oracle's keys bit for bit -- on a surface cloud, on exact ties
    (duplicates, lattice midpoints), on a cell far denser than stage A's 128 points, on a thin sheet, with near and far
    queries, on several cell sizes (regions that leave the grid on every side)."""
    rng = np.random.default_rng(32)
    try:
        gpu.set_nn_search(kernel)
        for name, (xyz, nrm) in _hard_cases().items():
            for sigma, cell in ((0.02, 0.0), (0.25, 0.0), (1.0, 0.0), (0.1, 0.6)):
                q = synth.queries(xyz, 6000, seed=7, sigma=sigma)
                q[:300] = xyz[rng.integers(0, xyz.shape[0], 300)].astype(np.float64)
                if name == "lattice":
                    q[300:900] = np.round(q[300:900] / 0.125) * 0.125      # midway between lattice points: exact ties
                c = gpu.Cloud(xyz, nrm, raw_lidar_frame=False, cell_size=cell)
                _check_exact(c.nn(q, gpu.NN_GRID), oracle.nn_bruteforce(xyz, q), f"kernel{kernel}/{name}/sigma{sigma}/cell{cell}")
                c.close()
    finally:
        gpu.set_nn_search(0)


@pytest.mark.parametrize("br", [(2, 1), (4, 1), (3, 2)])
def test_other_brick_geometries(gpu, oracle, br):
    """brick edges / halos other than the default 2 / 2 run on round 3's kernel (pcd_nn_set_tuning): still exact"""
    xyz, nrm = _clouds()["planes"]
    q = synth.queries(xyz, 6000, seed=9, sigma=0.3)
    try:
        gpu.set_nn_tuning(br[0], br[1], 0)
        c = gpu.Cloud(xyz, nrm, raw_lidar_frame=False)
        _check_exact(c.nn(q, gpu.NN_GRID), oracle.nn_bruteforce(xyz, q), f"brick{br}")
        c.close()
    finally:
        gpu.set_nn_tuning(2, 2, 0)


def test_clip_with_incoming_bounds(gpu, oracle):
    """the clip radius comes from the key a query arrives with when stage A finds nothing nearer: gate-bounded
    association (bound = gate) on the clipped kernel must accept exactly what the unbounded search + gate accept"""
    xyz, nrm = synth.cloud_planes(60000, seed=11, patches=12)
    q = synth.queries(xyz, 90000, seed=12, sigma=0.15)     # > 65536: the grid path
    c = gpu.Cloud(xyz, nrm, raw_lidar_frame=False)
    mr = np.full(q.shape[0], 0.2)
    a = c.associate(q, mr, gpu.GATE_MAPPER_LOCAL)
    b = c.associate(q, mr, gpu.GATE_MAPPER_LOCAL | gpu.GATE_BOUNDED_SEARCH)
    assert np.array_equal(a["type"], b["type"])
    acc = a["type"] != 0
    assert acc.any() and (~acc).any()
    assert np.array_equal(a["nn_idx"][acc], b["nn_idx"][acc])
    assert np.array_equal(a["abcd"][acc].view(np.uint64), b["abcd"][acc].view(np.uint64))
    c.close()


def test_radix_sort_bookkeeping_still_exact(gpu, oracle):
    """the round-2 bookkeeping (rocPRIM radix sort of (brick, query) pairs) stays as the path for grids with more than
    8 M occupied bricks: same keys as the counting-sort bookkeeping and the oracle"""
    xyz, nrm = synth.cloud_planes(60000, seed=4, patches=10)
    q = synth.queries(xyz, 9000, seed=5)
    q[::50] = np.nan
    q[1::50] += 500.0
    exp = oracle.nn_bruteforce(xyz, q)
    c = gpu.Cloud(xyz, nrm, raw_lidar_frame=False)
    try:
        for radix in (1, 0):
            gpu.set_nn_bookkeeping(radix)
            _check_exact(c.nn(q, gpu.NN_GRID), exp, f"bookkeeping radix={radix}")
    finally:
        gpu.set_nn_bookkeeping(0)
    c.close()


def test_large_sparse_grid_counting_sort(gpu):
    """a 40 M-cell grid (5 M bricks: 4096 fine keys per coarse bucket, the widest the counting-sort bookkeeping takes)
    against the device brute force"""
    rng = np.random.default_rng(31)
    n, Q = 200000, 20000
    xyz = (rng.random((n, 3)) * np.array([60.0, 60.0, 12.0])).astype(np.float32)
    nrm = np.zeros_like(xyz)
    nrm[:, 2] = 1.0
    q = (xyz[rng.integers(0, n, Q)] + rng.normal(0, 0.08, (Q, 3))).astype(np.float32)
    q[::97] = np.nan
    c = gpu.Cloud(xyz, nrm, raw_lidar_frame=False, cell_size=0.1)
    info = c.info()
    assert int(np.prod(info["dims"])) > 38_000_000, info
    exp = c.nn(q, gpu.NN_BRUTEFORCE)
    _check_exact(c.nn(q, gpu.NN_GRID), exp, "40 M-cell grid")
    c.close()
