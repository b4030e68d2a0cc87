"""GPU parity of the post-BA filter inputs against the oracle:
   base/reconstruction.cc:771-805 FilterLidarOutlier, base/projection.cc:104-117
   CalculateSquaredReprojectionError, base/reconstruction.cc:837-855 negative-depth test."""
import numpy as np
import pytest
import torch

from pcdhip import synth

pytestmark = pytest.mark.gpu


def test_observation_errors(gpu, oracle):
    s = synth.ba_scene(20, 5000, seed=41)
    s["poses"][:, :4] *= 1.7                    # un-normalised quaternions: the filter normalises first
    s["points"][::50, 2] += 80.0                # some points end up behind their cameras
    for model, cam in ((4, None), (2, [1200.0, 500.0, 400.0, 0.02]), (7, [900.0, 950.0, 500.0, 400.0, 0.8])):
        sc = dict(s)
        if cam is not None:
            sc["cam_model"] = np.array([model], np.int32)
            sc["cam_params_list"] = [cam]
        esq, edepth = oracle.BA(**sc).observation_errors()
        ba = gpu.BA(**sc)
        gsq, gdepth = ba.observation_errors()
        behind = edepth < np.finfo(np.float64).eps
        assert behind.any() and np.array_equal(gsq == np.finfo(np.float64).max, behind)
        np.testing.assert_allclose(gdepth, edepth, rtol=1e-12, atol=1e-12)
        ok = ~behind
        np.testing.assert_allclose(gsq[ok], esq[ok], rtol=1e-9, atol=1e-9)
        # decisions of the filters themselves
        assert np.array_equal(gdepth < np.finfo(np.float64).eps, behind)
        assert np.array_equal(gsq > 4.0 ** 2, esq > 4.0 ** 2)      # filter_max_reproj_error = 4 px
        ba.close()


def test_filter_lidar_outlier(gpu, oracle):
    rng = np.random.default_rng(8)
    n = 100_000
    X = rng.normal(size=(n, 3)) * 20
    lx = X + rng.normal(size=(n, 3)) * 0.6
    typ = rng.integers(0, 4, n).astype(np.uint8)            # 0 none, 1 Icp, 2 IcpGround, 3 Proj
    exp = oracle.filter_lidar_outlier(X, lx, typ, 1.5, 0.8)
    d = [torch.from_numpy(a).cuda() for a in (X, lx, typ)]
    out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    gpu.filter_lidar_outlier_device(d[0], d[1], d[2], n, 1.5, 0.8, out)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(got, exp)
    assert got[typ == 0].sum() == 0 and 0 < got.sum() < n


def test_filter_tracks_against_oracle(gpu, oracle):
    """per-track reduce of FilterPoints3DWithLargeReprojectionError / FilterObservationsWithNegativeDepth /
    ComputeMeanReprojectionError (base/reconstruction.cc:1662-1712, :837-855, :906-921) on a scene with short tracks,
    points behind cameras, unobserved points and large errors: every flag equal, errors and mean at 1e-12"""
    s = synth.ba_scene(12, 4000, seed=43)
    rng = np.random.default_rng(5)
    s["points"][::40, 2] += 80.0                          # behind their cameras
    s["obs_xy"][rng.integers(0, len(s["obs_xy"]), 600)] += rng.normal(0, 30, (600, 2))   # gross errors
    keep = np.ones(len(s["obs_point"]), bool)
    keep[np.isin(s["obs_point"], np.arange(0, 4000, 17))] = False    # points without observations
    first = np.unique(s["obs_point"], return_index=True)[1]
    only_one = np.isin(s["obs_point"], np.arange(5, 4000, 23))       # tracks of length 1
    keep &= ~only_one | np.isin(np.arange(len(keep)), first)
    for k in ("obs_image", "obs_point", "obs_xy"):
        s[k] = s[k][keep]
    ob = oracle.BA(**s)
    sq, depth = ob.observation_errors()
    fin = np.sqrt(sq[sq < 1e300])
    q70, q30 = float(np.quantile(fin, 0.7)), float(np.quantile(fin, 0.3))   # the poses are perturbed: errors of tens of px
    for max_err in (q70, q30, 0.0):
        exp = oracle.filter_tracks(sq, depth, s["obs_point"], 4000, max_err)
        ba = gpu.BA(**s)
        got = ba.filter_tracks(max_err)
        for k in ("obs_erase", "obs_negative_depth", "point_delete"):
            assert np.array_equal(got[k], exp[k]), (max_err, k)
        for k in ("num_filtered", "num_points_with_error", "num_negative_depth"):
            assert got[k] == exp[k], (max_err, k, got[k], exp[k])
        np.testing.assert_allclose(got["point_error"], exp["point_error"], rtol=1e-12, atol=1e-12)
        assert abs(got["mean_reproj_error"] - exp["mean_reproj_error"]) <= 1e-12 * max(1.0, exp["mean_reproj_error"])
        assert exp["num_negative_depth"] > 0
        if max_err == q70:
            assert 0 < exp["point_delete"].sum() < 4000 and 0 < exp["obs_erase"].sum() < len(exp["obs_erase"])
        ba.close()
