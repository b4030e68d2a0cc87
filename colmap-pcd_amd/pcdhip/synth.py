"""Seeded synthetic inputs of SURVEY.md section 8(d) (numpy; shared by tests/ and bench.py).

The Smith Hall dataset of the reference (README.md:61) is a Google-Drive link
and is not available offline, so clouds, queries and BA scenes are generated:
  cloud "planes"  : K rectangular planar patches in a 120 x 25 x 120 m box (visual frame: y is the
                    vertical axis, cf. the |ny/nx| > 10 ground test), points on a jittered 4 cm
                    lattice + N(0, 1 cm) along the normal, normals = patch normal + N(0, 0.02)
                    renormalised, 35 % of the patches horizontal.
  cloud "uniform" : U(box), random unit normals.
  queries         : random cloud points + N(0, 0.25 m) isotropic offset, 5 % replaced by U(box).
  BA scene        : cameras on a street-like path, OPENCV intrinsics fx = fy = 3039, 4032 x 3024,
                    tracks of length clip(Geom(0.25) + 2, 2, 30), observations = exact projection
                    + U(-2, 2) px, poses perturbed by N(0, 0.5 deg) / N(0, 5 cm).
Generators are numpy default_rng streams (the survey suggested std::mt19937_64; the
distributions and seeds are the same, the bit streams are not).
"""
import numpy as np

BOX = np.array([120.0, 25.0, 120.0])


def visual_to_raw(xyz, nrm):
    """Inverse of lidar/ply.cc:38-54: visual (x',y',z') = (-y,-z,x)  ->  raw (x,y,z) = (z',-x',-y')."""
    f = lambda a: np.stack([a[:, 2], -a[:, 0], -a[:, 1]], axis=1).astype(np.float32)
    return f(xyz), f(nrm)


def cloud_planes(n, seed=20240601, patches=256, spacing=0.04):
    rng = np.random.default_rng(seed)
    per = max(n // patches, 1)
    xyz = np.empty((n, 3), np.float32)
    nrm = np.empty((n, 3), np.float32)
    pos = 0
    for k in range(patches):
        m = per if k < patches - 1 else n - pos
        if m <= 0:
            break
        horizontal = rng.random() < 0.35
        if horizontal:
            nv = np.array([0.0, -1.0 if rng.random() < 0.5 else 1.0, 0.0])
            e1, e2 = np.array([1.0, 0, 0]), np.array([0, 0, 1.0])
        else:
            th = rng.random() * 2 * np.pi
            tilt = rng.normal(0, 0.05)
            nv = np.array([np.cos(th), tilt, np.sin(th)])
            nv /= np.linalg.norm(nv)
            e1 = np.cross(nv, [0, 1.0, 0]); e1 /= np.linalg.norm(e1)
            e2 = np.cross(nv, e1)
        aspect = np.exp(rng.uniform(-0.7, 0.7))
        na = max(int(np.sqrt(m * aspect)), 1)
        nb = (m + na - 1) // na
        c = rng.random(3) * BOX
        ij = np.arange(na * nb)[:m]
        u = (ij % na - na / 2) * spacing + rng.uniform(-0.01, 0.01, m)
        v = (ij // na - nb / 2) * spacing + rng.uniform(-0.01, 0.01, m)
        w = rng.normal(0, 0.01, m)
        p = c + u[:, None] * e1 + v[:, None] * e2 + w[:, None] * nv
        nn = nv + rng.normal(0, 0.02, (m, 3))
        nn /= np.linalg.norm(nn, axis=1, keepdims=True)
        xyz[pos:pos + m] = p
        nrm[pos:pos + m] = nn
        pos += m
    return xyz[:pos], nrm[:pos]


def cloud_uniform(n, seed=7, box=BOX):
    rng = np.random.default_rng(seed)
    xyz = (rng.random((n, 3)) * box).astype(np.float32)
    nn = rng.normal(size=(n, 3))
    nn /= np.linalg.norm(nn, axis=1, keepdims=True)
    return xyz, nn.astype(np.float32)


def queries(xyz, q, seed=99, sigma=0.25, outlier_frac=0.05, box=BOX):
    rng = np.random.default_rng(seed)
    pick = rng.integers(0, xyz.shape[0], q)
    pts = xyz[pick].astype(np.float64) + rng.normal(0, sigma, (q, 3))
    out = rng.random(q) < outlier_frac
    lo = xyz.min(axis=0).astype(np.float64)
    hi = xyz.max(axis=0).astype(np.float64)
    pts[out] = lo + rng.random((int(out.sum()), 3)) * (hi - lo)
    return pts


def max_range_schedule(q, seed=5):
    """per-point gate drawn from {1.5, 1.4, ..., 0.2} (sfm/incremental_mapper.cc:1159-1163)."""
    rng = np.random.default_rng(seed)
    return np.round(1.5 - 0.1 * rng.integers(0, 14, q), 1)


# --------------------------------------------------------------------- BA ---
def _quat_from_R(R):
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    x = np.sqrt(max(0.0, 1 + R[0, 0] - R[1, 1] - R[2, 2])) / 2
    y = np.sqrt(max(0.0, 1 - R[0, 0] + R[1, 1] - R[2, 2])) / 2
    z = np.sqrt(max(0.0, 1 - R[0, 0] - R[1, 1] + R[2, 2])) / 2
    x = np.copysign(x, R[2, 1] - R[1, 2]); y = np.copysign(y, R[0, 2] - R[2, 0]); z = np.copysign(z, R[1, 0] - R[0, 1])
    return np.array([w, x, y, z])


def _quat_rotate(q, p):
    w, v = q[..., :1], q[..., 1:]
    uv = 2 * np.cross(v, p)
    return p + w * uv + np.cross(v, uv)


def _opencv_project(params, u, v):
    fx, fy, cx, cy, k1, k2, p1, p2 = params
    u2, uv, v2 = u * u, u * v, v * v
    r2 = u2 + v2
    rad = k1 * r2 + k2 * r2 * r2
    du = u * rad + 2 * p1 * uv + p2 * (r2 + 2 * u2)
    dv = v * rad + 2 * p2 * uv + p1 * (r2 + 2 * v2)
    return fx * (u + du) + cx, fy * (v + dv) + cy


OPENCV_PARAMS = [3039.0, 3039.0, 2016.0, 1512.0, -0.05, 0.01, 1e-4, 1e-4]


def ba_scene(num_cams, num_points, seed=11, mean_extra=4.0, lidar_frac=0.9, const_pose_frac=0.0,
             max_track=30, scene_box=BOX, order="point", coherent=False):
    """Returns a dict of flat arrays for pcdhip.BA / oracle BA.

    Cameras sit on a path along x at y ~ 2 m, looking roughly along +z/-z; each point is observed by
    L cameras (L = clip(Geom(0.25)+2, 2, max_track)) chosen among those nearest in x; the observation is
    the exact projection + U(-2,2) px; poses are then perturbed.
    order = "point": observations grouped by track; "image": grouped by image (stable), the order in which
    BundleAdjuster::AddImageToProblem creates the residual blocks (optim/bundle_adjustment.cc:814-919).
    coherent = True: the point ids follow the images that see them (point ids ascend with the anchor image), the way an
    incremental reconstruction numbers its points -- every image triangulates new points as it is registered, so the
    points an image observes are neighbours in id space.  The default gives every point a RANDOM anchor image, the
    worst case for the per-observation point gathers."""
    rng = np.random.default_rng(seed)
    cams_x = np.linspace(5, scene_box[0] - 5, num_cams)
    poses_true = np.empty((num_cams, 7))
    centers = np.empty((num_cams, 3))
    for i in range(num_cams):
        yaw = rng.normal(0, 0.3) + (np.pi if i % 2 else 0.0)
        c, s = np.cos(yaw), np.sin(yaw)
        R_wc = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])      # camera-to-world
        C_w = np.array([cams_x[i], scene_box[1] * 0.5 + rng.normal(0, 0.2), scene_box[2] * 0.5 + rng.normal(0, 1.0)])
        R = R_wc.T
        t = -R @ C_w
        poses_true[i, :4] = _quat_from_R(R)
        poses_true[i, 4:] = t
        centers[i] = C_w
    # points in front of their cameras
    L = np.clip(rng.geometric(0.25, num_points) + 1, 2, min(max_track, num_cams))
    anchor = rng.integers(0, num_cams, num_points)
    obs_image, obs_point, obs_xy = [], [], []
    points = np.empty((num_points, 3))
    depth = rng.uniform(4, 30, num_points)
    lat = rng.uniform(-0.5, 0.5, num_points)
    ver = rng.uniform(-0.35, 0.35, num_points)
    qa = poses_true[anchor, :4]
    qinv = qa * np.array([1, -1, -1, -1])
    pc = np.stack([lat * depth, ver * depth, depth], axis=1)           # camera frame
    points[:] = _quat_rotate(qinv, pc - poses_true[anchor, 4:])        # world = R^T (pc - t)
    # neighbours with the same facing (i % 2) so the point is in front: a + 2*(j - L/2), j = 0..L-1
    half = (L // 2)
    obs_point = np.repeat(np.arange(num_points), L)
    j = np.arange(obs_point.shape[0]) - np.repeat(np.cumsum(L) - L, L)
    obs_image = anchor[obs_point] + 2 * (j - half[obs_point])
    inr = (obs_image >= 0) & (obs_image < num_cams)
    obs_image = obs_image[inr].astype(np.int32)
    obs_point = obs_point[inr].astype(np.int32)
    Pc = _quat_rotate(poses_true[obs_image, :4], points[obs_point]) + poses_true[obs_image, 4:]
    ok = Pc[:, 2] > 0.5
    obs_image, obs_point, Pc = obs_image[ok], obs_point[ok], Pc[ok]
    x, y = _opencv_project(OPENCV_PARAMS, Pc[:, 0] / Pc[:, 2], Pc[:, 1] / Pc[:, 2])
    # keep what a 4032 x 3024 image (plus a generous margin) can see: far off-axis rays leave the range in which the
    # distortion polynomial means anything and would dominate every norm in the tests
    vis = (np.abs(x - OPENCV_PARAMS[2]) < 4032.0) & (np.abs(y - OPENCV_PARAMS[3]) < 3024.0)
    obs_image, obs_point, x, y = obs_image[vis], obs_point[vis], x[vis], y[vis]
    obs_xy = np.stack([x, y], axis=1) + rng.uniform(-2, 2, (len(x), 2))
    if order == "image":
        perm = np.argsort(obs_image, kind="stable")
        obs_image, obs_point, obs_xy = obs_image[perm], obs_point[perm], obs_xy[perm]
    # perturb the poses (what BA starts from)
    poses = poses_true.copy()
    ang = rng.normal(0, np.deg2rad(0.5), (num_cams, 3))
    dq = np.concatenate([np.ones((num_cams, 1)), 0.5 * ang], axis=1)
    dq /= np.linalg.norm(dq, axis=1, keepdims=True)
    w1, v1 = dq[:, :1], dq[:, 1:]
    w2, v2 = poses[:, :1], poses[:, 1:4]
    poses[:, :4] = np.concatenate([w1 * w2 - np.sum(v1 * v2, 1, keepdims=True),
                                   w1 * v2 + w2 * v1 + np.cross(v1, v2)], axis=1)
    poses[:, 4:] += rng.normal(0, 0.05, (num_cams, 3))
    # lidar planes: random unit normals through a point near each 3D point
    nl = int(lidar_frac * num_points)
    lidar_point = np.sort(rng.choice(num_points, nl, replace=False)).astype(np.int32)
    nrm = rng.normal(size=(nl, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    lp = points[lidar_point] + rng.normal(0, 0.05, (nl, 3))
    abcd = np.concatenate([nrm, -np.sum(nrm * lp, 1, keepdims=True)], axis=1)
    ground = rng.random(nl) < 0.35
    weight = np.where(ground, 1000.0, 100.0)    # global-BA defaults, optim/bundle_adjustment.h:59-63
    const_pose = (rng.random(num_cams) < const_pose_frac).astype(np.uint8)
    if coherent:
        perm = np.argsort(anchor, kind="stable")              # new id -> old id
        new_id = np.empty(num_points, np.int64); new_id[perm] = np.arange(num_points)
        points = points[perm]
        obs_point = new_id[obs_point].astype(np.int32)
        if order == "point":
            o2 = np.argsort(obs_point, kind="stable")
            obs_image, obs_point, obs_xy = obs_image[o2], obs_point[o2], obs_xy[o2]
        lp_new = new_id[lidar_point]
        o3 = np.argsort(lp_new, kind="stable")
        lidar_point, abcd, weight = lp_new[o3].astype(np.int32), abcd[o3], weight[o3]
    return dict(cam_model=np.array([4], np.int32), cam_params_list=[OPENCV_PARAMS], poses=poses,
                image_camera=np.zeros(num_cams, np.int32), points=points, obs_image=obs_image,
                obs_point=obs_point, obs_xy=obs_xy, lidar_point=lidar_point, lidar_abcd=abcd,
                lidar_weight=weight, image_const_pose=const_pose)


# ------------------------------------------------------ depth projection ---
def proj_scene(num_images, feats_per_image, seed=3, scene_box=BOX, width=4032, height=3024, params=None,
               oob_frac=0.05):
    """Cameras inside the cloud box with random yaw / small pitch+roll (quaternions deliberately left a little
    un-normalised, the reference uses them as they are), OPENCV intrinsics, feature pixels U(image) with a few
    outside the image.  Returns (images, feat_xy) in the layout of pcdhip.Projector.set_new_images."""
    rng = np.random.default_rng(seed)
    params = list(OPENCV_PARAMS if params is None else params)
    images, feats = [], []
    pos = 0
    for i in range(num_images):
        yaw, pitch, roll = rng.uniform(0, 2 * np.pi), rng.normal(0, 0.1), rng.normal(0, 0.05)
        cy_, sy_ = np.cos(yaw), np.sin(yaw)
        cp, sp = np.cos(pitch), np.sin(pitch)
        cr, sr = np.cos(roll), np.sin(roll)
        Ry = np.array([[cy_, 0, sy_], [0, 1, 0], [-sy_, 0, cy_]])
        Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
        Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
        R_wc = Ry @ Rx @ Rz
        C_w = np.array([rng.uniform(10, scene_box[0] - 10), scene_box[1] * rng.uniform(0.3, 0.7),
                        rng.uniform(10, scene_box[2] - 10)])
        R = R_wc.T
        q = _quat_from_R(R) * (1.0 + rng.normal(0, 1e-3))
        t = -R @ C_w
        n = feats_per_image
        xy = np.stack([rng.uniform(0, width, n), rng.uniform(0, height, n)], axis=1)
        oob = rng.random(n) < oob_frac
        xy[oob] += rng.choice([-1.0, 1.0], (int(oob.sum()), 2)) * np.array([width, height]) * rng.uniform(0.0, 1.2)
        images.append(dict(qvec=q.tolist(), tvec=t.tolist(), params=params, width=width, height=height,
                           feat_begin=pos, feat_end=pos + n))
        feats.append(xy)
        pos += n
    return images, (np.concatenate(feats) if feats else np.zeros((0, 2)))
