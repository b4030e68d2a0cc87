"""Shared driver of the reference-held vectors in tests/golden/reference_kats.json (see its _source / _ref fields).

`backend` is anything with
  observation_errors(model, cam, qvec, tvec, points3D [n][3], obs [n][2]) -> (sq_err [n], depth [n])
  world_to_image(model, cam, uv [n][2]) -> xy [n][2]
so that tests/test_reference_kats_cpu.py (oracle) and tests/test_reference_kats_gpu.py (HIP, through the C ABI)
run the same checks.
"""
import json
import os

import numpy as np

KATS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kats.json")))
EPS = np.finfo(np.float64).eps


def check_squared_reprojection_error(backend):
    k = KATS["squared_reprojection_error"]
    X = np.array(k["points3D"], np.float64)
    exact = X[:, :2] / X[:, 2:3]                      # f = 1, c = 0, identity pose: the exact projection
    e0, _ = backend.observation_errors(k["model"], k["camera_params"], k["qvec"], k["tvec"], X, exact)
    assert (e0 == 0).all(), e0                        # BOOST_CHECK_EQUAL(error1, 0)
    e1, _ = backend.observation_errors(k["model"], k["camera_params"], k["qvec"], k["tvec"], X, exact + 1)
    np.testing.assert_allclose(e1, k["shifted_error"], rtol=k["shifted_rel_tol"])   # BOOST_CHECK_CLOSE(error3, 2, 1e-6)


def check_depth(backend):
    k = KATS["depth"]
    X = np.array([c["point3D"] for c in k["cases"]], np.float64)
    _, depth = backend.observation_errors(0, [1, 0, 0], k["qvec"], k["tvec"], X, np.zeros((len(X), 2)))
    np.testing.assert_allclose(depth, [c["depth"] for c in k["cases"]], rtol=1e-12, atol=0)
    # HasPointPositiveDepth: depth >= eps (base/projection.cc; what FilterObservationsWithNegativeDepth applies)
    assert list(depth >= EPS) == [c["positive"] for c in k["cases"]]


def check_quaternion_rotate_point(backend):
    """P = R(q / |q|) X is observed through the filter entry point: with tvec = (0,0,2) and SIMPLE_PINHOLE f = 1 the
    observation (Px, Py) / (Pz + 2) has squared error 0 and the depth is Pz + 2 -- all three components pinned."""
    for c in KATS["quaternion_rotate_point"]["cases"]:
        P = np.array(c["rotated"], np.float64)
        obs = (P[:2] / (P[2] + 2.0))[None, :]
        e, d = backend.observation_errors(0, [1, 0, 0], c["qvec"], [0, 0, 2], np.array([c["point"]], np.float64), obs)
        if c["exact"]:
            assert e[0] == 0 and d[0] == P[2] + 2.0, (c, e, d)
        else:
            assert e[0] < 1e-24 and abs(d[0] - (P[2] + 2.0)) < 1e-12, (c, e, d)


def _newton_inverse(forward, x0, y0, cam, model):
    """ImageToWorld by Newton iterations on WorldToImage (central-difference Jacobian), vectorised over points"""
    # start from the undistorted pinhole guess; focal / principal point positions as in base/camera_models.h
    f = np.array([cam[0], cam[0]]) if model in (0, 2, 3, 8, 9) else np.array([cam[0], cam[1]])
    c = np.array(cam[1:3]) if model in (0, 2, 3, 8, 9) else np.array(cam[2:4])
    tgt = np.stack([x0, y0], axis=1)
    uv = (tgt - c) / f
    h = 1e-7
    for _ in range(60):
        xy = forward(uv)
        r = xy - tgt
        if np.abs(r).max() < 1e-10:
            break
        du = np.array([h, 0.0]); dv = np.array([0.0, h])
        Ju = (forward(uv + du) - forward(uv - du)) / (2 * h)
        Jv = (forward(uv + dv) - forward(uv - dv)) / (2 * h)
        det = Ju[:, 0] * Jv[:, 1] - Jv[:, 0] * Ju[:, 1]
        step = np.stack([(Jv[:, 1] * r[:, 0] - Jv[:, 0] * r[:, 1]) / det,
                         (-Ju[:, 1] * r[:, 0] + Ju[:, 0] * r[:, 1]) / det], axis=1)
        uv = uv - step
    return uv


def check_camera_model_round_trips(backend, inverse_backend):
    """camera_models_test.cc:40-62,113-129: |ImageToWorld(WorldToImage(u0, v0)) - (u0, v0)| < 1e-6 on the world grid
    and |WorldToImage(ImageToWorld(x0, y0)) - (x0, y0)| < 1e-6 on the image grid + the principal point, for every
    parameter vector of the reference's tests.  WorldToImage = `backend` (the implementation under test);
    ImageToWorld = Newton inverse of `inverse_backend`'s WorldToImage (the oracle's), an independent evaluation."""
    k = KATS["camera_models"]
    tol = k["tolerance"]
    g = k["world_grid"]
    u = np.arange(g["lo"], g["hi"] + 1e-9, g["step"])
    U, V = [a.ravel() for a in np.meshgrid(u, u, indexing="ij")]
    gi = k["image_grid"]
    x = np.arange(gi["lo"], gi["hi"] + 1e-9, gi["step"])
    Xg, Yg = [a.ravel() for a in np.meshgrid(x, x, indexing="ij")]
    for case in k["cases"]:
        model, cam = case["model"], case["params"]
        fwd = lambda uv: backend.world_to_image(model, cam, uv)
        inv_fwd = lambda uv: inverse_backend.world_to_image(model, cam, uv)
        xy = fwd(np.stack([U, V], axis=1))
        assert np.isfinite(xy).all(), case
        uv_back = _newton_inverse(inv_fwd, xy[:, 0], xy[:, 1], cam, model)
        assert np.abs(uv_back - np.stack([U, V], axis=1)).max() < tol, (case, np.abs(uv_back[:, 0] - U).max())
        pp = cam[1:3] if model in (0, 2, 3, 8, 9) else cam[2:4]
        X0 = np.concatenate([Xg, [pp[0]]]); Y0 = np.concatenate([Yg, [pp[1]]])
        uv = _newton_inverse(inv_fwd, X0, Y0, cam, model)
        ok = np.isfinite(uv).all(axis=1)
        assert ok.all(), (case, "Newton inverse diverged")
        xy_back = fwd(uv)
        assert np.abs(xy_back - np.stack([X0, Y0], axis=1)).max() < tol, case


def check_filters(backend):
    """reconstruction_test.cc:394-445, :510-533, :599-614 through `backend.filter_tracks(model, cam, poses [I][7],
    points [P][3], obs_image, obs_point, obs_xy, max_reproj_error)` -> dict as pcdhip.BA.filter_tracks"""
    k = KATS["filter_points3d"]
    pose = list(k["qvec"]) + list(k["tvec"])
    for c in k["cases"]:
        n = len(c["track_images"])
        r = backend.filter_tracks(k["model"], k["camera_params"], [pose, pose], [c["point3D"]], list(range(n)), [0] * n,
                                  [k["obs"]] * n, c["max_reproj_error"])
        assert bool(r["point_delete"][0]) == c["deleted"], (c, r)
        assert r["num_filtered"] == (n if c["deleted"] else 0), (c, r)
        assert list(r["obs_erase"]) == [1 if c["deleted"] else 0] * n
        if not c["deleted"]:
            assert r["num_points_with_error"] == 1 and r["point_error"][0] >= 0
    # negative depth: point (0, 0, z) in image 1 (and a second image so that the track has two elements)
    for c in KATS["filter_negative_depth"]["cases"]:
        r = backend.filter_tracks(0, [1.0, 0.0, 0.0], [[1, 0, 0, 0, 0, 0, 0]] * 2, [[0.0, 0.0, c["z"]]], [0, 1], [0, 0],
                                  [[0.0, 0.0]] * 2, 1e300)
        assert list(r["obs_negative_depth"]) == [int(c["erased"])] * 2 and r["num_negative_depth"] == 2 * int(c["erased"]), (c, r)
    # mean reprojection error = mean of the surviving points' errors: points with a 2-element track whose elements
    # both miss by exactly e pixels get Error() = e
    for c in KATS["mean_reprojection_error"]["cases"]:
        errs = c["point_errors"]
        P = max(len(errs), 1)
        pts = [[0.0, 0.0, 1.0]] * P
        oi, op, oxy = [], [], []
        for p, e in enumerate(errs):
            oi += [0, 1]; op += [p, p]; oxy += [[e, 0.0], [0.0, e]]
        if not errs:                                   # no point has an error: a single-element track is deleted
            oi, op, oxy = [0], [0], [[0.0, 0.0]]
        r = backend.filter_tracks(0, [1.0, 0.0, 0.0], [[1, 0, 0, 0, 0, 0, 0]] * 2, pts, oi, op, oxy, 1e3)
        assert r["mean_reproj_error"] == c["mean"], (c, r)
