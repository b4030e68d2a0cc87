"""Closed-form checks of the depth-projection oracle (oracle/proj_oracle.c, restating lidar/pcd_projection.cc).

The reference has no tests or fixtures for this path ("parity unpinned"); these cases are hand-computed from the
reference's formulas: pinhole camera at the origin looking down +z, fx = fy = 3039, 4032 x 3024, scale 0.2."""
import numpy as np

PRM = [3039.0, 3039.0, 2016.0, 1512.0, 0.0, 0.0, 0.0, 0.0]


def _image(nf):
    return [dict(qvec=[1.0, 0, 0, 0], tvec=[0.0, 0, 0], params=PRM, width=4032, height=3024, feat_begin=0,
                 feat_end=nf)]


def _run(oracle, xyz, feat, **kw):
    oo = oracle.proj_options(**kw)
    c4 = oracle.proj_scale_coeffs(oo, PRM[0], PRM[1])
    xyz = np.asarray(xyz, np.float32).reshape(-1, 3)
    nrm = np.tile(np.array([0, 0, -1], np.float32), (xyz.shape[0], 1))
    feat = np.asarray(feat, np.float64).reshape(-1, 2)
    return oracle.proj_images(xyz, nrm, oo, c4, _image(feat.shape[0]), feat), c4


def test_scale_coeffs_quirk(oracle):
    oo = oracle.proj_options()
    a_x, b_x, a_y, b_y = oracle.proj_scale_coeffs(oo, 3039.0, 1519.5)
    assert np.isclose(a_x, (10 - 2) / (2 - 40.0)) and np.isclose(b_x, 2 - a_x * 40)
    assert np.isclose(a_y, (5 - 1) / (2 - 40.0))
    assert np.isclose(b_y, 2 - a_y * 40)          # pcd_projection.cc:397 uses the unscaled min_proj_scale


def test_splat_extent_and_nearest_wins(oracle):
    # point at depth 10 projects to (2016, 1512) -> scaled pixel (403, 302); half-width int(-8/38*10 + 2 + 320/38) = 8
    feats = [[(403 + 8) * 5 + 1, 1512], [(403 + 9) * 5 + 1, 1512], [2016, (302 - 8) * 5 + 1], [2016, (302 - 9) * 5 + 1]]
    (found, index, dist, l6, cam, pairs), _ = _run(oracle, [[0, 0, 10]], feats)
    assert found.tolist() == [1, 0, 1, 0] and pairs == 1
    assert dist[0] == np.float32(10) and index[0] == 0
    assert l6[0].tolist() == [0, 0, 10, 0, 0, -1]
    # ray/plane: plane z = 10 (normal (0,0,-1)), pixel ray through (u, v)
    u = feats[0][0]
    np.testing.assert_allclose(cam[0], [10 * (u - 2016) / 3039, 0, 10], rtol=1e-15)
    # a nearer point (depth 5, half-width int(-8/38*5 + 10.42) = 9) takes over the pixels its splat covers
    (found, index, dist, *_), _ = _run(oracle, [[0, 0, 10], [0, 0, 5]], feats)
    assert found.tolist() == [1, 1, 1, 1] and index.tolist() == [1, 1, 1, 1] and dist[0] == np.float32(5)
    # equal norms: the first point of the walk (same submap -> cloud order) stays
    (found, index, *_), _ = _run(oracle, [[0, 0, 10.25], [0, 0, 10.25]], feats[:1])
    assert index.tolist() == [0]
    # different submaps: key order (x, then y, then z) beats cloud order
    # (wide splats so that both points, 33 scaled pixels either side of the feature, cover it)
    (found, index, *_), _ = _run(oracle, [[0.55, 0, 10], [-0.55, 0, 10]], [[2016, 1512]], max_proj_scale=40,
                                 min_proj_scale=20)
    assert found[0] == 1 and index[0] == 1


def test_cull_is_on_submap_centres(oracle):
    # at depth 10 the image spans |x| < 10 * 2016 / 3039 = 6.63 m.  (6.4,0,10) lives in submap 6 (centre inside);
    # (6.55,0,10) projects inside the image too but lives in submap 7 whose centre is outside: never projected.
    def feat_of(x):
        return [[3039.0 * x / 10 + 2016, 1512]]
    (found, *_), _ = _run(oracle, [[6.4, 0, 10]], feat_of(6.4))
    assert found[0] == 1
    (found, _, _, _, _, pairs), _ = _run(oracle, [[6.55, 0, 10]], feat_of(6.55))
    assert found[0] == 0 and pairs == 0


def test_depth_gates(oracle):
    f = [[2016, 1512]]
    assert _run(oracle, [[0, 0, 0.4]], f, min_lidar_proj_dist=0.5)[0][0][0] == 0     # nearer than min_lidar_proj_dist
    assert _run(oracle, [[0, 0, 0.6]], f, min_lidar_proj_dist=0.5)[0][0][0] == 1
    assert _run(oracle, [[0, 0, -3]], f)[0][0][0] == 0                               # behind the camera
    assert _run(oracle, [[0, 0, 41.6]], f)[0][0][0] == 0                             # beyond choose_meter: culled
    # feature coordinates truncate toward zero: (-0.4 * 0.2) -> pixel 0, still inside the image
    # (point at the left image border: u0 = round((3039 * -0.65 + 2016) / 5) = 8, half-width 9)
    (found, *_), _ = _run(oracle, [[-1.43, 0, 2.2]], [[-0.4, 1512]], submap=0.1)
    assert found[0] == 1
    (found, *_), _ = _run(oracle, [[-1.43, 0, 2.2]], [[-5.1, 1512]], submap=0.1)     # pixel -1: outside
    assert found[0] == 0


def _golden():
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "proj_small.npz"))
    images = [dict(qvec=g["qvec"][i].tolist(), tvec=g["tvec"][i].tolist(), params=g["params"][i].tolist(),
                   width=int(g["size"][i, 0]), height=int(g["size"][i, 1]), feat_begin=int(g["feat_range"][i, 0]),
                   feat_end=int(g["feat_range"][i, 1])) for i in range(g["qvec"].shape[0])]
    return g, images


def test_oracle_reproduces_committed_fixture(oracle):
    g, images = _golden()
    oo = oracle.proj_options(min_lidar_proj_dist=float(g["min_lidar_proj_dist"]))
    found, index, dist, l6, cam, pairs = oracle.proj_images(g["xyz"], g["nrm"], oo, g["coeffs"], images, g["feat"])
    assert np.array_equal(found, g["found"]) and np.array_equal(index, g["index"])
    assert np.array_equal(dist.view(np.uint32), g["dist_bits"]) and pairs == int(g["pairs"])
    assert np.array_equal(cam, g["cam_xyz"])
