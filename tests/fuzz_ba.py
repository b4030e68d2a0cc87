"""(test infrastructure, run by hand; not collected by pytest)  Randomised parity sweep of the BA evaluation against the Jet oracle: python tests/fuzz_ba.py [seconds] [seed].
Random scene sizes (incl. segments > 1024 observations per image, images without observations, tracks of length 1..max,
points without observations), camera models (one model per scene or mixed), several cameras, constant poses / tvec
masks / points, lidar terms, the three losses, observation order (by track / by image / shuffled), refined-camera
masks.  Compared: cost, H_img, g_img, H_pt, g_pt, W (fused and raw), raw residual / Jacobian blocks, camera blocks."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colmap-pcd_amd")); sys.path.insert(0, ROOT)
import numpy as np
import pcdhip
from pcdhip import synth
from oracle import pyoracle as oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
NP = [3, 4, 4, 5, 8, 8, 12, 5, 4, 5, 12]


def close(a, b, rtol, what, ctx):
    a = np.asarray(a); b = np.asarray(b)
    scale = max(1.0, float(np.abs(b).max()) if b.size else 1.0)
    if not np.allclose(a, b, rtol=rtol, atol=rtol * scale, equal_nan=True):
        bad = np.nonzero(~np.isclose(a, b, rtol=rtol, atol=rtol * scale, equal_nan=True))
        print("MISMATCH", what, ctx, "count", len(bad[0]), "first", [x[:3] for x in bad], a[bad][:3], b[bad][:3], flush=True)
        sys.exit(1)


def cam_params(model):
    # plausible parameters for every model: f, (f2), cx, cy, small distortion
    K = NP[model]
    p = np.zeros(K)
    two_f = model in (1, 4, 5, 6, 10)
    p[0] = rng.uniform(900, 1500)
    i = 1
    if two_f:
        p[1] = p[0] * rng.uniform(0.98, 1.02); i = 2
    p[i] = rng.uniform(1900, 2100); p[i + 1] = rng.uniform(1400, 1600)
    p[i + 2:] = rng.normal(0, 5e-3, K - i - 2)
    if model == 7:   # FOV: omega
        p[4] = rng.uniform(0.2, 0.9)
    return p


ncase = 0
while time.time() < t_end:
    I = int(rng.choice([1, 2, 5, 17, 40]))
    P = int(rng.choice([1, 30, 500, 4000]))
    if rng.random() < 0.15:
        I, P = 2, 9000           # > 1024 observations per image: several segments
    s = synth.ba_scene(I, P, seed=int(rng.integers(1 << 30)), const_pose_frac=float(rng.choice([0, 0.25, 1.0])),
                       order=str(rng.choice(["point", "image"])))
    O = len(s["obs_image"])
    if O and rng.random() < 0.3:   # shuffled observation order
        perm = rng.permutation(O)
        s["obs_image"], s["obs_point"], s["obs_xy"] = s["obs_image"][perm], s["obs_point"][perm], s["obs_xy"][perm]
    mixed = rng.random() < 0.3
    C = int(rng.integers(1, 4))
    models = [int(rng.integers(0, 11)) for _ in range(C)] if mixed else [int(rng.integers(0, 11))] * C
    s["cam_model"] = np.array(models, np.int32)
    s["cam_params_list"] = [cam_params(m) for m in models]
    s["image_camera"] = rng.integers(0, C, I).astype(np.int32)
    tv = rng.integers(0, 8, I).astype(np.uint8) * (rng.random(I) < 0.3)
    pc = (rng.random(P) < rng.choice([0.0, 0.1, 1.0])).astype(np.uint8)
    loss = [(0, 1.0), (1, 1.0), (2, 2.5), (1, 0.3)][int(rng.integers(0, 4))]
    kw = dict(image_const_tvec=tv.astype(np.uint8), point_const=pc, loss_type=loss[0], loss_scale=loss[1])
    ctx = dict(I=I, P=P, O=O, L=len(s.get("lidar_point", [])), models=models, loss=loss, case=ncase)
    ob = oracle.BA(**s, **kw)
    ba = pcdhip.BA(**s, **kw)
    cost, Himg, gimg, Hpt, gpt, W = ob.normal_equations(want_w=True)
    got = ba.evaluate(("cost", "H_img", "g_img", "H_pt", "g_pt", "W"))
    if abs(got["cost"][0] - cost) > 1e-10 * max(abs(cost), 1e-30):
        print("MISMATCH cost", ctx, got["cost"][0], cost); sys.exit(1)
    for k, ref in (("H_img", Himg), ("g_img", gimg), ("H_pt", Hpt), ("g_pt", gpt), ("W", W)):
        close(got[k], ref, 1e-8, k, ctx)
    close(ba.evaluate(("W",))["W"], W, 1e-8, "W raw", ctx)
    c1 = ba.evaluate(("cost",))["cost"][0]
    if abs(c1 - cost) > 1e-10 * max(abs(cost), 1e-30):
        print("MISMATCH cost-only pass", ctx, c1, cost); sys.exit(1)
    res, Jq, Jt, JX, Jc, JL = ob.evaluate_raw()
    raw = ba.evaluate(("residuals", "jac_q", "jac_t", "jac_X", "jac_lidar", "jac_cam"))
    close(raw["residuals"], res, 1e-9, "residuals", ctx)
    close(raw["jac_q"], Jq, 1e-8, "jac_q", ctx); close(raw["jac_t"], Jt, 1e-8, "jac_t", ctx)
    close(raw["jac_X"], JX, 1e-8, "jac_X", ctx); close(raw["jac_lidar"], JL, 1e-11, "jac_lidar", ctx)
    if O:
        Kmax = max(NP[m] for m in models)
        close(raw["jac_cam"][:, :, :Kmax], Jc[:, :, :Kmax], 1e-8, "jac_cam", ctx)
    ba.close()
    if rng.random() < 0.5 and O:
        flags = [bool(rng.integers(0, 2)) for _ in range(3)]
        const_cams = tuple(int(c) for c in range(C) if rng.random() < 0.3)
        mask = pcdhip.camera_refine_mask(s["cam_model"], *flags, constant_cameras=const_cams)
        H, g, E, Wc = ob.camera_blocks(mask, want_w=True)
        bc = pcdhip.BA(**s, **kw, camera_refine=mask)
        gc = bc.evaluate(("H_cam", "g_cam", "E_cam", "W_cam"))
        for k, ref in (("H_cam", H), ("g_cam", g), ("E_cam", E), ("W_cam", Wc)):
            close(gc[k], ref, 1e-8, k, dict(ctx, flags=flags, const_cams=const_cams))
        bc.close()
    ncase += 1
    if ncase % 20 == 0:
        print("cases %d, %.0f s left" % (ncase, t_end - time.time()), flush=True)
print("OK: %d cases, no mismatch" % ncase)
