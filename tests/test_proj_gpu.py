"""GPU parity of the depth-projection association (lidar/pcd_projection.cc) against the oracle.

The winner per feature pixel is index work: bit-exact cloud row and float norm.  The two read-outs are copies
(6-vector) and a handful of double operations (ray/plane), compared to 1e-12 relative."""
import numpy as np
import pytest

from pcdhip import synth

pytestmark = pytest.mark.gpu


def _run(gpu, oracle, xyz, nrm, images, feat, **opt_kw):
    oo = oracle.proj_options(**opt_kw)
    coeffs = oracle.proj_scale_coeffs(oo, images[0]["params"][0], images[0]["params"][1])
    exp = oracle.proj_images(xyz, nrm, oo, coeffs, images, feat)
    cloud = gpu.Cloud(xyz, nrm, raw_lidar_frame=False)
    kw = dict(depth_image_scale=oo.depth_image_scale, max_proj_scale=oo.max_proj_scale,
              min_proj_scale=oo.min_proj_scale, min_proj_dist=oo.min_proj_dist, submap_length=oo.submap_length,
              submap_width=oo.submap_width, submap_height=oo.submap_height, choose_meter=oo.choose_meter,
              min_lidar_proj_dist=oo.min_lidar_proj_dist)
    pj = gpu.Projector(cloud, **kw)
    got = pj.set_new_images(images, feat)
    c4, latched = pj.scale_coeffs()
    assert latched and np.array_equal(c4, coeffs)
    assert pj.last_pairs == exp[5]
    pj.close()
    cloud.close()
    return got, exp


def _compare(got, exp):
    found, index, dist, l6, cam = got
    efound, eindex, edist, el6, ecam, _ = exp
    assert np.array_equal(found, efound)
    assert np.array_equal(index, eindex)
    assert np.array_equal(dist.view(np.uint32), edist.view(np.uint32))
    assert np.array_equal(l6, el6)
    np.testing.assert_allclose(cam, ecam, rtol=1e-12, atol=0)


def test_proj_planes(gpu, oracle):
    xyz, nrm = synth.cloud_planes(400_000, seed=5)
    images, feat = synth.proj_scene(6, 3000, seed=2)
    got, exp = _run(gpu, oracle, xyz, nrm, images, feat)
    assert 0.02 < exp[0].mean() < 0.98          # both outcomes occur
    _compare(got, exp)


def test_proj_uniform_cloud_other_options(gpu, oracle):
    xyz, nrm = synth.cloud_uniform(300_000, seed=9)
    prm = [1500.0, 1480.0, 1000.0, 760.0, 0.08, -0.02, 3e-4, -2e-4]
    images, feat = synth.proj_scene(4, 2000, seed=7, width=2000, height=1500, params=prm)
    got, exp = _run(gpu, oracle, xyz, nrm, images, feat, depth_image_scale=0.25, max_proj_scale=6, min_proj_scale=1,
                    min_proj_dist=3.0, submap=2.0, choose_meter=25.0, min_lidar_proj_dist=1.0)
    assert exp[0].any()
    _compare(got, exp)


def test_proj_ties_take_first_in_walk_order(gpu, oracle):
    # duplicated points: equal norms, the winner must be the first one the reference's walk meets
    xyz, nrm = synth.cloud_planes(60_000, seed=1)
    xyz = np.concatenate([xyz, xyz[::-1]]).copy()
    nrm = np.concatenate([nrm, -nrm[::-1]]).copy()
    images, feat = synth.proj_scene(3, 2000, seed=4)
    got, exp = _run(gpu, oracle, xyz, nrm, images, feat)
    assert exp[0].any()
    _compare(got, exp)


def test_proj_edge_cases(gpu, oracle):
    xyz, nrm = synth.cloud_planes(50_000, seed=3)
    xyz[7] = [np.inf, 0, 0]                       # non-finite row: never a candidate
    images, feat = synth.proj_scene(3, 500, seed=8)
    images[1]["feat_end"] = images[1]["feat_begin"]              # image without features
    images[2]["width"], images[2]["height"] = 3, 3               # scaled image of 0 x 0 pixels
    got, exp = _run(gpu, oracle, xyz, nrm, images, feat)
    _compare(got, exp)
    assert not got[0][images[1]["feat_begin"]:].any()
    # empty batch / empty cloud are fine
    cloud = gpu.Cloud(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), raw_lidar_frame=False)
    pj = gpu.Projector(cloud)
    assert pj.num_submaps == 0
    f, idx, d, l6, cam = pj.set_new_images(images[:1], feat)
    assert not f.any()
    pj.close()
    cloud.close()


def test_proj_coeffs_latch_on_first_camera(gpu, oracle):
    # the reference's function-local statics: a second camera with another focal length reuses the first's slope
    xyz, nrm = synth.cloud_planes(200_000, seed=5)
    im1, f1 = synth.proj_scene(1, 2000, seed=2)
    im2, f2 = synth.proj_scene(1, 2000, seed=6, params=[2000.0, 2100.0, 2016.0, 1512.0, 0, 0, 0, 0])
    oo = oracle.proj_options()
    c_first = oracle.proj_scale_coeffs(oo, 3039.0, 3039.0)
    exp2 = oracle.proj_images(xyz, nrm, oo, c_first, im2, f2)
    cloud = gpu.Cloud(xyz, nrm, raw_lidar_frame=False)
    pj = gpu.Projector(cloud, min_lidar_proj_dist=0.5)
    pj.set_new_images(im1, f1)
    got2 = pj.set_new_images(im2, f2)
    _compare(got2, exp2)
    pj.close()
    cloud.close()


def test_proj_golden_fixture(gpu):
    """committed vectors (tests/golden/proj_small.npz, generator make_golden.py): no oracle needed at run time"""
    from tests.test_proj_cpu import _golden
    g, images = _golden()
    cloud = gpu.Cloud(g["xyz"], g["nrm"], raw_lidar_frame=False)
    pj = gpu.Projector(cloud, min_lidar_proj_dist=float(g["min_lidar_proj_dist"]))
    found, index, dist, l6, cam = pj.set_new_images(images, g["feat"])
    assert np.array_equal(found, g["found"]) and np.array_equal(index, g["index"])
    assert np.array_equal(dist.view(np.uint32), g["dist_bits"])
    np.testing.assert_allclose(cam, g["cam_xyz"], rtol=1e-12, atol=0)
    assert pj.last_pairs == int(g["pairs"])
    pj.close()
    cloud.close()
