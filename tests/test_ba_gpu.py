"""GPU parity of the bundle-adjustment evaluation (through the C ABI) against the Jet oracle.

Reference: base/cost_functions.h:49-141, :150-241, :256-370; base/camera_models.h (11 models);
optim/bundle_adjustment.cc:53-68 (loss), :694-1131 (block structure, constness).
Bar (north_star): residuals within 1e-6 relative; tested here at 1e-9 relative on residuals and
Jacobians (scale-aware absolute floor), 1e-8 on the accumulated normal-equation blocks.
"""
import json
import os

import numpy as np
import pytest

from pcdhip import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _close(a, b, rtol, what):
    scale = max(1.0, float(np.abs(b).max()) if b.size else 1.0)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol * scale, err_msg=what)


def _scene(model, rng, I=5, P=120):
    """random multi-model scene: every image gets camera `model`"""
    from tests.test_oracle_cpu import _rand_block
    q, t, X, cam, obs = _rand_block(rng, model)
    s = synth.ba_scene(I, P, seed=int(rng.integers(1 << 30)), const_pose_frac=0.25)
    # re-project the scene's observations with this camera model through the oracle so residuals stay small
    s["cam_model"] = np.array([model], np.int32)
    s["cam_params_list"] = [cam]
    return s


@pytest.mark.parametrize("model", range(11))
def test_raw_blocks_all_camera_models(gpu, oracle, model):
    rng = np.random.default_rng(200 + model)
    s = _scene(model, rng)
    ob = oracle.BA(**s)
    res, Jq, Jt, JX, Jc, JL = ob.evaluate_raw()
    ba = gpu.BA(**s)
    got = ba.evaluate(("residuals", "jac_q", "jac_t", "jac_X", "jac_lidar", "jac_cam"))
    _close(got["residuals"], res, 1e-9, "residuals")
    # camera-parameter block (refine_focal_length / principal_point / extra_params): 2 x K in a stride-12 row
    K = oracle.lib().oracle_camera_num_params(model)
    assert Jc.shape[2] >= K and np.abs(Jc[:, :, :K]).max() > 0
    _close(got["jac_cam"][:, :, :K], Jc[:, :, :K], 1e-9, "Jc")
    assert not got["jac_cam"][:, :, K:].any()
    _close(got["jac_q"], Jq, 1e-9, "Jq")
    _close(got["jac_t"], Jt, 1e-9, "Jt")
    _close(got["jac_X"], JX, 1e-9, "JX")
    _close(got["jac_lidar"], JL, 1e-12, "JL")
    ba.close()


def test_reference_known_answers_on_gpu(gpu):
    """src/base/cost_functions_test.cc:41-99: exact residuals for both functors"""
    k = json.load(open(os.path.join(HERE, "golden", "cost_function_kats.json")))
    for cpose in (0, 1):
        for c in k["cases"]:
            ba = gpu.BA([k["model"]], [c["camera_params"]], [c["qvec"] + c["tvec"]], [0], [c["point3D"]], [0], [0],
                        [c["obs"]], image_const_pose=[cpose])
            r = ba.evaluate(("residuals",))["residuals"]
            assert list(r) == c["residuals"]
            ba.close()


@pytest.mark.parametrize("loss", [(0, 1.0), (1, 1.0), (2, 2.5)])
def test_normal_equations_and_cost(gpu, oracle, loss):
    s = synth.ba_scene(12, 3000, seed=21, const_pose_frac=0.25)
    I, P = 12, 3000
    tv = np.zeros(I, np.uint8); tv[1] = 0b001; tv[2] = 0b110
    pc = np.zeros(P, np.uint8); pc[::13] = 1
    kw = dict(image_const_tvec=tv, point_const=pc, loss_type=loss[0], loss_scale=loss[1])
    ob = oracle.BA(**s, **kw)
    cost, Himg, gimg, Hpt, gpt, W = ob.normal_equations(want_w=True)
    ba = gpu.BA(**s, **kw)
    got = ba.evaluate(("cost", "H_img", "g_img", "H_pt", "g_pt", "W"))
    assert abs(got["cost"][0] - cost) <= 1e-11 * abs(cost)
    _close(got["H_img"], Himg, 1e-9, "H_img")
    _close(got["g_img"], gimg, 1e-9, "g_img")
    _close(got["H_pt"], Hpt, 1e-9, "H_pt")
    _close(got["g_pt"], gpt, 1e-9, "g_pt")
    _close(got["W"], W, 1e-9, "W")
    # W alone comes from the raw kernel instead of riding on the image pass: same blocks
    _close(ba.evaluate(("W",))["W"], got["W"], 1e-13, "W raw vs fused")
    # observations in image-major order (AddImageToProblem's): the fused pass writes W[o] through its e -> o map
    perm = np.argsort(s["obs_image"], kind="stable")
    s_img = dict(s); s_img["obs_image"], s_img["obs_point"], s_img["obs_xy"] = s["obs_image"][perm], s["obs_point"][perm], s["obs_xy"][perm]
    ba_i = gpu.BA(**s_img, **kw)
    got_i = ba_i.evaluate(("cost", "H_img", "W"))
    _close(got_i["W"], W[perm], 1e-9, "W image-major")
    _close(got_i["H_img"], Himg, 1e-9, "H_img image-major")
    ba_i.close()
    # constant poses / points contribute no blocks
    cp = s["image_const_pose"].astype(bool)
    assert cp.any() and not got["H_img"][cp].any() and not got["H_pt"][pc.astype(bool)].any()
    # determinism: bitwise identical on re-evaluation (fixed-order reductions, no atomics)
    again = ba.evaluate(("cost", "H_img", "g_img", "H_pt"))
    assert again["cost"][0] == got["cost"][0] and np.array_equal(again["H_img"], got["H_img"])
    # the cost-only pass (LM trial step) sums per observation instead of per track: same value up to the
    # summation order, and bitwise reproducible itself
    c1 = ba.evaluate(("cost",))["cost"][0]
    c2 = ba.evaluate(("cost", "H_img"))["cost"][0]
    assert c1 == c2 and abs(c1 - cost) <= 1e-11 * abs(cost)
    # parameter update path
    poses2 = s["poses"].copy(); poses2[:, 4:] += 0.01
    ba.set_parameters(poses=poses2)
    s2 = dict(s); s2["poses"] = poses2
    c2 = oracle.BA(**s2, **kw).normal_equations()[0]
    assert abs(ba.evaluate(("cost",))["cost"][0] - c2) <= 1e-11 * abs(c2)
    ba.close()


def test_lidar_term_on_plane_and_errors(gpu):
    # s == 0: residual 0, Jacobian 0 (sign(0) = 0; Ceres' autodiff would give NaN)
    ba = gpu.BA([0], [[1.0, 0, 0]], [[1, 0, 0, 0, 0, 0, 0]], [0], [[0.3, 0.5, 4.0]], [0], [0], [[0.0, 0.0]],
                lidar_point=[0], lidar_abcd=[[0, 1, 0, -0.5]], lidar_weight=[100.0])
    out = ba.evaluate(("residuals", "jac_lidar"))
    assert out["residuals"][2] == 0 and not out["jac_lidar"].any()
    ba.close()
    with pytest.raises(gpu.PcdError):
        gpu.BA([99], [[1.0, 0, 0]], [[1, 0, 0, 0, 0, 0, 0]], [0], [[0, 0, 1.0]], [0], [0], [[0.0, 0.0]])
    with pytest.raises(gpu.PcdError):
        gpu.BA([0], [[1.0, 0, 0]], [[1, 0, 0, 0, 0, 0, 0]], [0], [[0, 0, 1.0]], [5], [0], [[0.0, 0.0]])


def test_midsize_scene_properties(gpu, oracle):
    """200 cameras / 100 k points: cost equals the host sum of the returned residuals; gradient blocks
    equal J^T r re-assembled from the returned raw blocks on a sample."""
    s = synth.ba_scene(200, 100_000, seed=31)
    ba = gpu.BA(**s)
    out = ba.evaluate(("cost", "residuals", "jac_X", "jac_lidar", "g_pt"))
    res = out["residuals"]
    assert abs(out["cost"][0] - 0.5 * float(res @ res)) <= 1e-10 * out["cost"][0]
    O = len(s["obs_image"])
    g = np.zeros((100_000, 3))
    np.add.at(g, s["obs_point"], np.einsum("ork,or->ok", out["jac_X"], res[:2 * O].reshape(-1, 2)))
    np.add.at(g, s["lidar_point"], out["jac_lidar"] * res[2 * O:, None])
    _close(out["g_pt"], g, 1e-9, "g_pt vs J^T r")
    ob = oracle.BA(**s)
    _close(res, ob.evaluate_raw()[0], 1e-9, "residuals vs oracle")
    ba.close()


@pytest.mark.parametrize("model", [2, 4, 6, 9])
def test_camera_blocks(gpu, oracle, model):
    """refined intrinsics (ParameterizeCameras, optim/bundle_adjustment.cc:1047-1100): camera blocks of the normal
    equations vs the oracle's accumulation of its Jet Jacobians, with the focal / principal-point / extra-parameter
    subsets of bundle_adjustment_test.cc:479-645, a constant camera, constant poses / tvec entries / points and a
    robust loss.  Models 2 and 4 take the compiled-in kernels, 6 and 9 the generic (per-observation switch) path."""
    rng = np.random.default_rng(300 + model)
    s = _scene(model, rng, I=7, P=400)
    # three cameras of this model with different parameters; camera 1 constant (SetConstantCamera)
    cam = np.array(s["cam_params_list"][0], np.float64)
    s["cam_model"] = np.array([model] * 3, np.int32)
    s["cam_params_list"] = [cam, cam * (1 + 1e-3), cam * (1 - 2e-3)]
    s["image_camera"] = (np.arange(7) % 3).astype(np.int32)
    tv = np.zeros(7, np.uint8); tv[3] = 0b001
    pc = np.zeros(400, np.uint8); pc[::7] = 1
    kw = dict(image_const_tvec=tv, point_const=pc, loss_type=1, loss_scale=2.0)
    for flags in [(True, False, True), (False, False, True), (True, True, True), (True, False, False)]:
        mask = gpu.camera_refine_mask(s["cam_model"], *flags, constant_cameras=(1,))
        ob = oracle.BA(**s, **kw)
        H, g, E, W = ob.camera_blocks(mask, want_w=True)
        ba = gpu.BA(**s, **kw, camera_refine=mask)
        got = ba.evaluate(("H_cam", "g_cam", "E_cam", "W_cam", "H_img", "cost"))
        _close(got["H_cam"], H, 1e-9, f"H_cam {flags}")
        _close(got["g_cam"], g, 1e-9, f"g_cam {flags}")
        _close(got["E_cam"], E, 1e-9, f"E_cam {flags}")
        _close(got["W_cam"], W, 1e-9, f"W_cam {flags}")
        assert np.abs(H[0]).max() > 0 and not got["H_cam"][1].any() and not got["E_cam"][1::3].any()
        # constant parameters have empty rows / columns
        const = np.nonzero(mask[:len(cam)] == 0)[0]
        assert not got["H_cam"][0][const].any() and not got["H_cam"][0][:, const].any()
        again = ba.evaluate(("H_cam", "g_cam"))
        assert np.array_equal(again["H_cam"], got["H_cam"]) and np.array_equal(again["g_cam"], got["g_cam"])
        # the pose blocks do not depend on the camera mask
        _close(got["H_img"], ob.normal_equations()[1], 1e-9, "H_img")
        ba.close()
    # no mask: intrinsics constant (this fork's default), camera outputs zero
    ba = gpu.BA(**s, **kw)
    got = ba.evaluate(("H_cam", "g_cam", "E_cam"))
    assert not got["H_cam"].any() and not got["g_cam"].any() and not got["E_cam"].any()
    ba.close()


def _pose_tangent_gradient(s, out, res):
    """g_img re-assembled on the host from the raw blocks: sum over the image's observations of
    [Jq plus(q) | Jt]^T r (trivial loss, no constant poses in these scenes)"""
    O = len(s["obs_image"])
    x = s["poses"][s["obs_image"], :4]
    plus = np.stack([np.stack([-x[:, 1], -x[:, 2], -x[:, 3]], 1), np.stack([x[:, 0], x[:, 3], -x[:, 2]], 1),
                     np.stack([-x[:, 3], x[:, 0], x[:, 1]], 1), np.stack([x[:, 2], -x[:, 1], x[:, 0]], 1)], 1)  # [O][4][3]
    Jp = np.concatenate([np.einsum("ork,okc->orc", out["jac_q"], plus), out["jac_t"]], axis=2)               # [O][2][6]
    g = np.zeros((s["poses"].shape[0], 6))
    np.add.at(g, s["obs_image"], np.einsum("orc,or->oc", Jp, res[:2 * O].reshape(-1, 2)))
    return g


@pytest.mark.parametrize("cams,points,shard", [(450, 400_000, None), (5000, 2_000_000, 8), (5000, 2_000_000, None)],
                         ids=["config B: 450 cams / 400k pts / ~2M obs", "config C: one rank's share of 5000 cams / 2M tracks",
                              "config C at full size on one GPU: 5000 cams / 2M tracks / ~9.5M obs"])
def test_config_size_properties(gpu, oracle, cams, points, shard):
    """BASELINE.json's BA sizes -- B (Smith Hall 450: 450 cameras / 400 k points / ~2 M observations / ~350 k LiDAR terms),
    one rank's eighth of C (5 000 cameras / 2 M tracks / ~10 M observations, sharded by track as bench.py does) and C
    whole on one GPU (it fits: 2.9 ms per step, profiles/r03_bench_config_C_single_gpu.json) --
    checked through properties that do not depend on the size: cost = 1/2 |r|^2 of the returned residuals, point and
    pose gradients = J^T r re-assembled on the host from the returned raw blocks, W = Jp^T JX on a sample, and the
    oracle's Jet residuals / Jacobians on a track sample."""
    from pcdhip import dist as pd
    s = synth.ba_scene(cams, points, seed=41, lidar_frac=0.875, order="image")
    if shard:
        s, _ = pd.shard_tracks(s, 3, shard)
    s["image_const_pose"] = np.zeros(cams, np.uint8)
    P, O, L = s["points"].shape[0], len(s["obs_image"]), len(s["lidar_point"])
    assert (P, cams) == ((points // shard if shard else points), s["poses"].shape[0]) and O > 4 * P and L > 0.8 * P
    ba = gpu.BA(**s)
    out = ba.evaluate(("cost", "residuals", "jac_q", "jac_t", "jac_X", "jac_lidar", "g_pt", "g_img", "H_pt", "W"))
    res = out["residuals"]
    assert np.isfinite(res).all()
    assert abs(out["cost"][0] - 0.5 * float(res @ res)) <= 1e-10 * out["cost"][0]
    r2 = res[:2 * O].reshape(-1, 2)
    g = np.zeros((P, 3))
    np.add.at(g, s["obs_point"], np.einsum("ork,or->ok", out["jac_X"], r2))
    np.add.at(g, s["lidar_point"], out["jac_lidar"] * res[2 * O:, None])
    _close(out["g_pt"], g, 1e-9, "g_pt vs J^T r")
    _close(out["g_img"], _pose_tangent_gradient(s, out, res), 1e-9, "g_img vs J^T r")
    H = np.zeros((P, 3, 3))
    np.add.at(H, s["obs_point"], np.einsum("ora,orb->oab", out["jac_X"], out["jac_X"]))
    np.add.at(H, s["lidar_point"], np.einsum("la,lb->lab", out["jac_lidar"], out["jac_lidar"]))
    _close(out["H_pt"], H, 1e-9, "H_pt vs J^T J")
    # W on a sample of observations
    sel = np.random.default_rng(5).choice(O, 20000, replace=False)
    x = s["poses"][s["obs_image"][sel], :4]
    plus = np.stack([np.stack([-x[:, 1], -x[:, 2], -x[:, 3]], 1), np.stack([x[:, 0], x[:, 3], -x[:, 2]], 1),
                     np.stack([-x[:, 3], x[:, 0], x[:, 1]], 1), np.stack([x[:, 2], -x[:, 1], x[:, 0]], 1)], 1)
    Jp = np.concatenate([np.einsum("ork,okc->orc", out["jac_q"][sel], plus), out["jac_t"][sel]], axis=2)
    _close(out["W"][sel], np.einsum("ora,orb->oab", Jp, out["jac_X"][sel]), 1e-9, "W vs Jp^T JX")
    # oracle (Jets) on the tracks of the first 4000 points
    keep = s["obs_point"] < 4000
    lk = s["lidar_point"] < 4000
    sub = dict(s)
    sub["points"] = s["points"][:4000]
    sub["obs_image"], sub["obs_point"], sub["obs_xy"] = s["obs_image"][keep], s["obs_point"][keep], s["obs_xy"][keep]
    sub["lidar_point"], sub["lidar_abcd"], sub["lidar_weight"] = s["lidar_point"][lk], s["lidar_abcd"][lk], s["lidar_weight"][lk]
    ores, oJq, oJt, oJX, _, oJL = oracle.BA(**sub).evaluate_raw()
    _close(res[:2 * O].reshape(-1, 2)[keep].ravel(), ores[:2 * int(keep.sum())], 1e-9, "residuals vs oracle")
    _close(out["jac_q"][keep], oJq, 1e-9, "Jq vs oracle")
    _close(out["jac_X"][keep], oJX, 1e-9, "JX vs oracle")
    _close(res[2 * O:][lk], ores[2 * int(keep.sum()):], 1e-9, "lidar residuals vs oracle")
    ba.close()


def test_sparse_scenes_cost_partials(gpu, oracle):
    """Scenes with far fewer residual blocks than points (local BA with one observation per point; lidar-only):
    k_ba_points runs one workgroup per two 64-track slices whatever O + L is, and each workgroup writes one cost
    partial -- round 2 sized that buffer from O + L and wrote past its end.  The cost must equal the
    oracle's and the bytes behind the partials (the next allocation, the handle's cost scalar buffer) must survive."""
    P = 100_000
    rng = np.random.default_rng(77)
    s = synth.ba_scene(6, P, seed=78)
    # (a) one observation per point: keep the first observation of every track
    first = np.unique(s["obs_point"], return_index=True)[1]
    a = dict(s)
    a["obs_image"], a["obs_point"], a["obs_xy"] = s["obs_image"][first], s["obs_point"][first], s["obs_xy"][first]
    keep = rng.random(len(s["lidar_point"])) < 0.02
    a["lidar_point"], a["lidar_abcd"], a["lidar_weight"] = s["lidar_point"][keep], s["lidar_abcd"][keep], s["lidar_weight"][keep]
    assert len(a["obs_image"]) + keep.sum() < 2 * P
    # (b) no observations at all, a few lidar terms
    b = dict(a)
    b["obs_image"], b["obs_point"], b["obs_xy"] = a["obs_image"][:0], a["obs_point"][:0], a["obs_xy"][:0]
    for sc in (a, b):
        cost = oracle.BA(**sc).normal_equations()[0]
        ba = gpu.BA(**sc)
        for _ in range(2):
            got = ba.evaluate(("cost", "H_pt", "g_pt"))
            assert abs(got["cost"][0] - cost) <= 1e-11 * max(abs(cost), 1e-300)
            assert abs(ba.evaluate(("cost",))["cost"][0] - cost) <= 1e-11 * max(abs(cost), 1e-300)
        ba.close()
