"""CPU tests that pin the oracle (no GPU): reference known answers, independent re-derivations
(scipy cKDTree, finite differences, torch float64 autograd), committed golden vectors."""
import json
import os

import numpy as np
import pytest

from pcdhip import synth

HERE = os.path.dirname(os.path.abspath(__file__))


# ---------------------------------------------------------------- NN ------
def test_bruteforce_equals_kdtree_and_scipy(oracle):
    from scipy.spatial import cKDTree
    xyz, _ = synth.cloud_planes(30000, seed=5, patches=10)
    xyz[1000:1100] = xyz[0:100]
    q = synth.queries(xyz, 2500, seed=6, sigma=0.4)
    q[:50] = xyz[1000:1050].astype(np.float64)
    i1, d1, f1 = oracle.nn_bruteforce(xyz, q)
    i2, d2, f2 = oracle.KDTree(xyz).query(q)
    assert np.array_equal(i1, i2) and np.array_equal(d1.view(np.uint32), d2.view(np.uint32)) and np.array_equal(f1, f2)
    assert (i1[:50] == np.arange(50)).all()          # ties -> lowest index
    # independent implementation on float32-representable inputs: same distances, same index unless tied
    dd, ii = cKDTree(xyz.astype(np.float64)).query(q.astype(np.float32).astype(np.float64))
    np.testing.assert_allclose(dd ** 2, d1, rtol=2e-6, atol=1e-12)
    diff = ii != i1
    assert diff.mean() < 0.05
    for j in np.nonzero(diff)[0]:                     # every disagreement is a (near-)tie
        a = np.sum((xyz[ii[j]].astype(np.float64) - q[j].astype(np.float32)) ** 2)
        assert abs(a - d1[j]) <= 4e-6 * max(d1[j], 1e-12) + 1e-12


def test_nn_edge_semantics(oracle):
    xyz = np.array([[np.inf, 0, 0], [1, 2, 3], [1, 2, 3]], np.float32)
    q = np.array([[1.0, 2, 3], [np.nan, 0, 0], [1e30, 0, 0]])
    i, d, f = oracle.nn_bruteforce(xyz, q)
    assert list(f) == [1, 0, 0] and i[0] == 1 and d[0] == 0
    i2, d2, f2 = oracle.KDTree(xyz).query(q)
    assert np.array_equal(i, i2) and np.array_equal(f, f2)
    e = np.zeros((0, 3), np.float32)
    assert not oracle.nn_bruteforce(e, q)[2].any() and not oracle.KDTree(e).query(q)[2].any()


def test_direction_trans(oracle):
    # lidar/ply.cc:38-54
    raw = np.array([[1, 2, 3], [4, np.nan, 6], [7, 8, 9]], np.float32)
    nr = np.array([[0, 0, 1], [0, 1, 0], [np.nan, 0, 0]], np.float32)
    x, n = oracle.direction_trans(raw, nr)
    assert x.shape == (1, 3) and list(x[0]) == [-2, -3, 1] and list(n[0]) == [0, -1, 0]
    xv, nv = synth.cloud_uniform(100, seed=1)
    rx, rn = synth.visual_to_raw(xv, nv)
    bx, bn = oracle.direction_trans(rx, rn)
    assert np.array_equal(bx, xv) and np.array_equal(bn, nv)


def test_golden_nn_and_assoc(oracle):
    g = np.load(os.path.join(HERE, "golden", "nn_small.npz"))
    a = np.load(os.path.join(HERE, "golden", "assoc_small.npz"))
    i, d, f = oracle.nn_bruteforce(g["xyz"], g["q"])
    assert np.array_equal(i, g["idx"]) and np.array_equal(d.view(np.uint32), g["sqdist_bits"]) and np.array_equal(f, g["found"])
    out6, ok = oracle.search_nearest_neibor(g["xyz"], g["nrm"], i, f)
    assert np.array_equal(out6, a["out6"]) and np.array_equal(ok, a["ok"])
    assert ok[42] == 0 and ok[40] == 0                # zero normal / NaN query rejected
    for mode in (0, 1, 2):
        abcd, typ, dist, ang, d2p = oracle.associate(g["q"], out6, ok, None if mode == 2 else a["max_range"], mode)
        assert np.array_equal(typ, a[f"type{mode}"]) and np.array_equal(abcd, a[f"abcd{mode}"])
        assert np.array_equal(dist, a[f"dist{mode}"]) and np.array_equal(ang, a[f"angle{mode}"], equal_nan=True)  # 0/0 angle when X == winner, as in the reference


# ------------------------------------------------------- association ------
def test_assoc_closed_forms(oracle):
    # hand-computed: plane y = 0.5 through (1,0.5,2), normal (0,2,0) -> n^ = (0,1,0), d = -0.5
    X = np.array([[1.3, 0.9, 2.4]])
    l6 = np.array([[1.0, 0.5, 2.0, 0.0, 2.0, 0.0]])
    ok = np.array([1], np.uint8)
    abcd, typ, dist, ang, d2p = oracle.associate(X, l6, ok, 1.5, 0)
    assert np.allclose(abcd[0], [0, 1, 0, -0.5]) and typ[0] == 2            # |ny/nx| = inf > 10 -> ground
    p2p = np.sqrt(0.3 ** 2 + 0.4 ** 2 + 0.4 ** 2)
    assert abs(dist[0] - p2p) < 1e-15 and abs(d2p[0] - 0.4) < 1e-15 and abs(ang[0] - 0.4 / p2p) < 1e-15
    # gates: mapper drops beyond max_range, controller drops dist2plane > 1 or p2p > 2
    assert oracle.associate(X, l6, ok, 0.5, 0)[1][0] == 0
    assert oracle.associate(X, l6, ok, None, 2)[1][0] == 2
    far = np.array([[1.0, 2.0, 2.0]])
    assert oracle.associate(far, l6, ok, None, 2)[1][0] == 0                 # dist2plane 1.5 > 1
    # classification uses the raw normal with IEEE division: 0/0 -> NaN -> Icp
    l6b = np.array([[1.0, 0.5, 2.0, 0.0, 0.0, 1.0]])
    assert oracle.associate(X, l6b, ok, 1.5, 0)[1][0] == 1
    l6c = np.array([[1.0, 0.5, 2.0, 0.05, 1.0, 0.05]])
    assert oracle.associate(X, l6c, ok, 1.5, 0)[1][0] == 2
    l6d = np.array([[1.0, 0.5, 2.0, 0.2, 1.0, 0.05]])
    assert oracle.associate(X, l6d, ok, 1.5, 0)[1][0] == 1
    assert oracle.associate(X, l6, np.array([0], np.uint8), 1.5, 0)[1][0] == 0


# ---------------------------------------------------------------- BA ------
def test_reference_known_answers(oracle):
    """src/base/cost_functions_test.cc:41-99 (exact BOOST_CHECK_EQUAL in the reference)"""
    k = json.load(open(os.path.join(HERE, "golden", "cost_function_kats.json")))
    for c in k["cases"]:
        r = oracle.reproj_residual(k["model"], c["qvec"], c["tvec"], c["point3D"], c["camera_params"], c["obs"])
        assert list(r) == c["residuals"]
        r2 = oracle.reproj_block(k["model"], c["qvec"], c["tvec"], c["point3D"], c["camera_params"], c["obs"])[0]
        assert list(r2) == c["residuals"]


def test_lidar_block_closed_form(oracle):
    # X=(0.3,-0.2,4), abcd=(0,1,0,-0.5), w=100 -> r=70, J=(0,-100,0)
    r, J = oracle.lidar_block([0.3, -0.2, 4], [0, 1, 0, -0.5], 100)
    assert r == 70.0 and list(J) == [0.0, -100.0, 0.0]
    r, J = oracle.lidar_block([0.3, 0.5, 4], [0, 1, 0, -0.5], 100)        # exactly on the plane
    assert r == 0 and list(J) == [0, 0, 0]
    r, J = oracle.lidar_block([0.3, 0.5, 4], [0, 1, 0, -0.5], 100, strict=True)
    assert r == 0 and np.isnan(J).all()                                     # Ceres' Jet of sqrt at 0


def _rand_block(rng, model):
    K = [3, 4, 4, 5, 8, 8, 12, 5, 4, 5, 12][model]
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    t = rng.normal(size=3)
    Pc = np.array([rng.uniform(-0.3, 0.3) * 6, rng.uniform(-0.3, 0.3) * 6, rng.uniform(4, 9)])
    # world point so that R X + t = Pc
    w, v = q[0], q[1:]
    qi = np.concatenate([[w], -v])
    p = Pc - t
    uv = 2 * np.cross(qi[1:], p)
    X = p + qi[0] * uv + np.cross(qi[1:], uv)
    f = rng.uniform(800, 1500)
    base = {3: [f, 500, 400], 4: [f, 1.1 * f, 500, 400]}
    if model == 0: cam = base[3]
    elif model == 1: cam = base[4]
    elif model in (2, 8): cam = [f, 500, 400, rng.uniform(-0.1, 0.1)]
    elif model in (3, 9): cam = [f, 500, 400, rng.uniform(-0.1, 0.1), rng.uniform(-0.02, 0.02)]
    elif model in (4, 5): cam = base[4] + list(rng.uniform(-0.05, 0.05, 4))
    elif model == 7: cam = base[4] + [rng.uniform(0.3, 1.2)]
    else: cam = base[4] + list(rng.uniform(-0.02, 0.02, 8))
    assert len(cam) == K
    obs = rng.uniform(0, 1000, 2)
    return q, t, X, np.array(cam), obs


@pytest.mark.parametrize("model", range(11))
def test_jet_jacobians_vs_finite_differences(oracle, model):
    rng = np.random.default_rng(100 + model)
    for _ in range(5):
        q, t, X, cam, obs = _rand_block(rng, model)
        r, Jq, Jt, JX, Jc = oracle.reproj_block(model, q, t, X, cam, obs)
        # Jet division is f.a * (1/g.a) (as in ceres/jet.h), plain double is f/g: equal to a few ulp
        assert np.allclose(r, oracle.reproj_residual(model, q, t, X, cam, obs), rtol=1e-13, atol=1e-10)
        def fd(arg, n):
            J = np.zeros((2, n))
            for k in range(n):
                h = 1e-6 * max(1.0, abs([q, t, X, cam][arg][k]))
                a = [q.copy(), t.copy(), X.copy(), cam.copy()]; b = [q.copy(), t.copy(), X.copy(), cam.copy()]
                a[arg][k] += h; b[arg][k] -= h
                J[:, k] = (oracle.reproj_residual(model, *a, obs) - oracle.reproj_residual(model, *b, obs)) / (2 * h)
            return J
        for J, arg, n in ((Jq, 0, 4), (Jt, 1, 3), (JX, 2, 3), (Jc, 3, len(cam))):
            ref = fd(arg, n)
            assert np.allclose(J, ref, rtol=2e-5, atol=2e-5 * max(1.0, np.abs(ref).max())), (model, arg, J, ref)


def test_opencv_block_vs_torch_autograd(oracle):
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(7)
    for _ in range(4):
        q, t, X, cam, obs = _rand_block(rng, 4)
        def f(qv, tv, Xv, cv):
            w, v = qv[0], qv[1:]
            uv = 2 * torch.linalg.cross(v, Xv)
            P = Xv + w * uv + torch.linalg.cross(v, uv) + tv
            u, vv = P[0] / P[2], P[1] / P[2]
            fx, fy, cx, cy, k1, k2, p1, p2 = cv
            u2, uvv, v2 = u * u, u * vv, vv * vv
            r2 = u2 + v2
            rad = k1 * r2 + k2 * r2 * r2
            du = u * rad + 2 * p1 * uvv + p2 * (r2 + 2 * u2)
            dv = vv * rad + 2 * p2 * uvv + p1 * (r2 + 2 * v2)
            return torch.stack([fx * (u + du) + cx - obs[0], fy * (vv + dv) + cy - obs[1]])
        args = [torch.tensor(a, dtype=torch.float64) for a in (q, t, X, cam)]
        J = torch.autograd.functional.jacobian(f, tuple(args))
        r, Jq, Jt, JX, Jc = oracle.reproj_block(4, q, t, X, cam, obs)
        np.testing.assert_allclose(r, f(*args).numpy(), rtol=1e-12, atol=1e-9)
        for got, ref in zip((Jq, Jt, JX, Jc), J):
            np.testing.assert_allclose(got, ref.numpy(), rtol=1e-9, atol=1e-8)


def test_losses(oracle):
    # Ceres loss_function.cc: rho(s), rho'(s), rho''(s)
    for s in (0.0, 0.3, 4.0, 1e3):
        assert list(oracle.loss(0, 1.0, s)) == [s, 1.0, 0.0]
        a = 1.7
        r = oracle.loss(1, a, s)
        assert np.allclose(r, [2 * a * a * (np.sqrt(1 + s / a / a) - 1), 1 / np.sqrt(1 + s / a / a),
                               -1 / (2 * a * a * (1 + s / a / a) ** 1.5)])
        r = oracle.loss(2, a, s)
        assert np.allclose(r, [a * a * np.log(1 + s / a / a), 1 / (1 + s / a / a), -1 / (a * a * (1 + s / a / a) ** 2)])
        assert oracle.loss(1, a, s)[2] <= 0 and oracle.loss(2, a, s)[2] <= 0      # Corrector: sqrt(rho') branch


def test_normal_equations_consistency(oracle):
    """H = J^T J, g = J^T r assembled from the raw blocks + loss + manifold, re-derived with numpy."""
    s = synth.ba_scene(6, 150, seed=5, const_pose_frac=0.3)
    tv = np.zeros(6, np.uint8); tv[1] = 0b001
    pc = np.zeros(150, np.uint8); pc[::11] = 1
    for loss in ((0, 1.0), (1, 1.0), (2, 2.0)):
        ba = oracle.BA(**s, image_const_tvec=tv, point_const=pc, loss_type=loss[0], loss_scale=loss[1])
        res, Jq, Jt, JX, Jc, JL = ba.evaluate_raw()
        cost, Himg, gimg, Hpt, gpt, W = ba.normal_equations(want_w=True)
        O = len(ba.obs_image)
        H2 = np.zeros_like(Himg); g2 = np.zeros_like(gimg); P2 = np.zeros_like(Hpt); p2 = np.zeros_like(gpt); c2 = 0.0
        for o in range(O):
            im, pt = ba.obs_image[o], ba.obs_point[o]
            r = res[2 * o:2 * o + 2]
            rho = oracle.loss(loss[0], loss[1], r @ r)
            c2 += 0.5 * rho[0]; sr = np.sqrt(rho[1])
            x = ba.poses[im, :4]
            plus = np.array([[-x[1], -x[2], -x[3]], [x[0], x[3], -x[2]], [-x[3], x[0], x[1]], [x[2], -x[1], x[0]]])
            Jp = np.concatenate([Jq[o] @ plus, Jt[o]], axis=1) * sr
            if ba.image_const_pose[im]: Jp[:] = 0
            for k in range(3):
                if (tv[im] >> k) & 1: Jp[:, 3 + k] = 0
            Jx = JX[o] * sr * (0 if pc[pt] else 1)
            H2[im] += Jp.T @ Jp; g2[im] += Jp.T @ (sr * r); P2[pt] += Jx.T @ Jx; p2[pt] += Jx.T @ (sr * r)
            np.testing.assert_allclose(W[o], Jp.T @ Jx, rtol=1e-12, atol=1e-9)
        for l in range(len(ba.lidar_point)):
            pt = ba.lidar_point[l]; r = res[2 * O + l]
            rho = oracle.loss(loss[0], loss[1], r * r)
            c2 += 0.5 * rho[0]; sr = np.sqrt(rho[1])
            if not pc[pt]:
                J = JL[l] * sr
                P2[pt] += np.outer(J, J); p2[pt] += J * sr * r
        assert abs(cost - c2) <= 1e-12 * abs(c2)
        for a, b in ((Himg, H2), (gimg, g2), (Hpt, P2), (gpt, p2)):
            np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-7)


def test_camera_blocks_consistency(oracle):
    """camera blocks (refined intrinsics, optim/bundle_adjustment.cc:1047-1100) = Jc^T [Jc | Jp | JX | r] re-assembled
    with numpy from the raw Jet blocks, the loss and the parameter mask."""
    s = synth.ba_scene(5, 120, seed=9, const_pose_frac=0.2)
    s["cam_model"] = np.array([4, 4], np.int32)
    s["cam_params_list"] = [synth.OPENCV_PARAMS, [3000.0, 3010.0, 2000.0, 1500.0, -0.04, 0.02, 2e-4, -1e-4]]
    s["image_camera"] = np.array([0, 1, 0, 1, 0], np.int32)
    tv = np.zeros(5, np.uint8); tv[2] = 0b100
    pc = np.zeros(120, np.uint8); pc[::9] = 1
    ba = oracle.BA(**s, image_const_tvec=tv, point_const=pc, loss_type=2, loss_scale=3.0)
    mask = np.array([1, 1, 0, 0, 1, 1, 1, 1] + [0] * 8, np.uint8)      # camera 0: f + extra; camera 1 constant
    H, g, E, W = ba.camera_blocks(mask, want_w=True)
    res, Jq, Jt, JX, Jc, JL = ba.evaluate_raw()
    H2 = np.zeros_like(H); g2 = np.zeros_like(g); E2 = np.zeros_like(E)
    for o in range(len(ba.obs_image)):
        im, pt = ba.obs_image[o], ba.obs_point[o]
        cm = ba.image_camera[im]
        r = res[2 * o:2 * o + 2]
        rho = oracle.loss(2, 3.0, r @ r)
        sr = np.sqrt(rho[1])
        x = ba.poses[im, :4]
        plus = np.array([[-x[1], -x[2], -x[3]], [x[0], x[3], -x[2]], [-x[3], x[0], x[1]], [x[2], -x[1], x[0]]])
        Jp = np.concatenate([Jq[o] @ plus, Jt[o]], axis=1) * sr
        if ba.image_const_pose[im]: Jp[:] = 0
        for k in range(3):
            if (tv[im] >> k) & 1: Jp[:, 3 + k] = 0
        Jx = JX[o] * sr * (0 if pc[pt] else 1)
        J = np.zeros((2, 12)); J[:, :8] = Jc[o][:, :8] * sr * mask[8 * cm:8 * cm + 8]
        H2[cm] += J.T @ J; g2[cm] += J.T @ (sr * r); E2[im] += J.T @ Jp
        np.testing.assert_allclose(W[o], J.T @ Jx, rtol=1e-12, atol=1e-6)
    for a, b in ((H, H2), (g, g2), (E, E2)):
        np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-6)
    assert np.abs(H[0]).max() > 0 and not H[1].any() and not H[0][2:4].any()
