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
