"""ctypes loader for oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  The product path (colmap-pcd_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build():
    """Compile the oracle with gcc/g++ (no GPU, no reference sources needed)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


class BAProblem(C.Structure):
    _fields_ = [
        ("num_cameras", C.c_int32), ("cam_model", C.c_void_p), ("cam_param_off", C.c_void_p),
        ("cam_params", C.c_void_p),
        ("num_images", C.c_int32), ("poses", C.c_void_p), ("image_camera", C.c_void_p),
        ("image_const_pose", C.c_void_p), ("image_const_tvec", C.c_void_p),
        ("num_points", C.c_int32), ("points", C.c_void_p), ("point_const", C.c_void_p),
        ("num_obs", C.c_int64), ("obs_image", C.c_void_p), ("obs_point", C.c_void_p), ("obs_xy", C.c_void_p),
        ("num_lidar", C.c_int64), ("lidar_point", C.c_void_p), ("lidar_abcd", C.c_void_p),
        ("lidar_weight", C.c_void_p),
        ("loss_type", C.c_int32), ("loss_scale", C.c_double), ("cam_jac_stride", C.c_int32),
    ]


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    L.oracle_direction_trans.restype = C.c_uint64
    L.oracle_direction_trans.argtypes = [f32p, f32p, C.c_uint64, f32p, f32p]
    L.oracle_nn_bruteforce.restype = None
    L.oracle_nn_bruteforce.argtypes = [f32p, C.c_uint64, f64p, C.c_uint64, u32p, f32p, u8p]
    L.oracle_kdtree_build.restype = C.c_void_p
    L.oracle_kdtree_build.argtypes = [f32p, C.c_uint64, C.c_int]
    L.oracle_kdtree_free.restype = None
    L.oracle_kdtree_free.argtypes = [C.c_void_p]
    L.oracle_kdtree_query.restype = None
    L.oracle_kdtree_query.argtypes = [C.c_void_p, f64p, C.c_uint64, u32p, f32p, u8p]
    L.oracle_search_nearest_neibor.restype = None
    L.oracle_search_nearest_neibor.argtypes = [f32p, f32p, u32p, u8p, C.c_uint64, f64p, u8p]
    L.oracle_associate.restype = None
    L.oracle_associate.argtypes = [f64p, f64p, u8p, C.c_void_p, C.c_uint64, C.c_int, f64p, u8p, f64p, f64p, f64p]
    L.oracle_search_range_schedule.restype = None
    L.oracle_search_range_schedule.argtypes = [i32p, C.c_uint64, C.c_double, C.c_double, C.c_double, f64p]
    L.oracle_camera_num_params.restype = C.c_int
    L.oracle_camera_num_params.argtypes = [C.c_int]
    L.oracle_reproj_residual.restype = None
    L.oracle_reproj_residual.argtypes = [C.c_int, f64p, f64p, f64p, f64p, f64p, f64p]
    L.oracle_reproj_block.restype = None
    L.oracle_reproj_block.argtypes = [C.c_int, f64p, f64p, f64p, f64p, f64p, f64p, f64p, f64p, f64p, f64p]
    L.oracle_lidar_block.restype = None
    L.oracle_lidar_block.argtypes = [f64p, f64p, C.c_double, C.c_int, f64p, f64p]
    L.oracle_loss.restype = None
    L.oracle_loss.argtypes = [C.c_int, C.c_double, C.c_double, f64p]
    L.oracle_ba_evaluate_raw.restype = None
    L.oracle_ba_evaluate_raw.argtypes = [C.POINTER(BAProblem)] + [C.c_void_p] * 6
    L.oracle_ba_residuals.restype = None
    L.oracle_ba_residuals.argtypes = [C.POINTER(BAProblem), C.c_void_p]
    L.oracle_ba_normal_equations.restype = C.c_double
    L.oracle_ba_normal_equations.argtypes = [C.POINTER(BAProblem)] + [C.c_void_p] * 5
    L.oracle_sift_match.restype = C.c_int
    L.oracle_sift_match.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_float, C.c_float, C.c_int, u32p, i32p, i32p]
    L.oracle_sift_distance_matrix.restype = None
    L.oracle_sift_distance_matrix.argtypes = [u8p, C.c_int, u8p, C.c_int, i32p]
    L.oracle_sift_random_descriptors.restype = None
    L.oracle_sift_random_descriptors.argtypes = [C.c_int, u8p]
    L.oracle_sift_renormalize_row.restype = None
    L.oracle_sift_renormalize_row.argtypes = [f32p, u8p]
    _LIB = L
    return L


# ---------------------------------------------------------------- SIFT ----
def sift_match(d1, d2, max_ratio=0.8, max_distance=0.7, cross_check=True):
    """MatchSiftFeaturesCPUBruteForce: returns (matches [M][2] uint32, m12 [n1] int32, m21 [n2] int32)"""
    d1 = np.ascontiguousarray(d1, np.uint8).reshape(-1, 128)
    d2 = np.ascontiguousarray(d2, np.uint8).reshape(-1, 128)
    n1, n2 = d1.shape[0], d2.shape[0]
    m = np.zeros((max(n1, 1), 2), np.uint32)
    m12 = np.full(max(n1, 1), -1, np.int32)
    m21 = np.full(max(n2, 1), -1, np.int32)
    n = lib().oracle_sift_match(d1.reshape(-1) if n1 else np.zeros(1, np.uint8), n1,
                                d2.reshape(-1) if n2 else np.zeros(1, np.uint8), n2,
                                max_ratio, max_distance, int(cross_check), m.reshape(-1), m12, m21)
    return m[:n].copy(), m12[:n1], m21[:n2]


def sift_distance_matrix(d1, d2):
    d1 = np.ascontiguousarray(d1, np.uint8).reshape(-1, 128)
    d2 = np.ascontiguousarray(d2, np.uint8).reshape(-1, 128)
    out = np.empty((d1.shape[0], d2.shape[0]), np.int32)
    lib().oracle_sift_distance_matrix(d1.reshape(-1), d1.shape[0], d2.reshape(-1), d2.shape[0], out.reshape(-1))
    return out


def sift_random_descriptors(n):
    """CreateRandomFeatureDescriptors of src/feature/sift_test.cc:243-253 (mt19937 seed 0)"""
    out = np.zeros((max(n, 1), 128), np.uint8)
    if n:
        lib().oracle_sift_random_descriptors(n, out.reshape(-1))
    return out[:n].copy()


def sift_renormalize_row(row_u8):
    out = np.zeros(128, np.uint8)
    lib().oracle_sift_renormalize_row(np.ascontiguousarray(row_u8, np.float32), out)
    return out


# ------------------------------------------------------------------ NN ----
def direction_trans(xyz_raw, nrm_raw):
    xyz_raw = np.ascontiguousarray(xyz_raw, np.float32)
    nrm_raw = np.ascontiguousarray(nrm_raw, np.float32)
    n = xyz_raw.shape[0]
    xo = np.empty((n, 3), np.float32)
    no = np.empty((n, 3), np.float32)
    m = lib().oracle_direction_trans(xyz_raw, nrm_raw, n, xo, no)
    return xo[:m].copy(), no[:m].copy()


def nn_bruteforce(xyz, q):
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    q = np.ascontiguousarray(q, np.float64).reshape(-1, 3)
    nq = q.shape[0]
    idx = np.empty(nq, np.uint32)
    sq = np.empty(nq, np.float32)
    found = np.empty(nq, np.uint8)
    lib().oracle_nn_bruteforce(xyz, xyz.shape[0], q, nq, idx, sq, found)
    return idx, sq, found


class KDTree:
    def __init__(self, xyz, leaf_max=15):
        self._xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        self._h = lib().oracle_kdtree_build(self._xyz, self._xyz.shape[0], leaf_max)

    def query(self, q):
        q = np.ascontiguousarray(q, np.float64).reshape(-1, 3)
        nq = q.shape[0]
        idx = np.empty(nq, np.uint32)
        sq = np.empty(nq, np.float32)
        found = np.empty(nq, np.uint8)
        lib().oracle_kdtree_query(self._h, q, nq, idx, sq, found)
        return idx, sq, found

    def query_mt(self, q, nthreads):
        """the same serial search, queries split over `nthreads` host threads (the tree is read-only; ctypes
        releases the GIL).  bench.py's all-cores CPU baseline; the reference itself queries serially."""
        from concurrent.futures import ThreadPoolExecutor
        q = np.ascontiguousarray(q, np.float64).reshape(-1, 3)
        nq = q.shape[0]
        idx = np.empty(nq, np.uint32)
        sq = np.empty(nq, np.float32)
        found = np.empty(nq, np.uint8)
        cuts = np.linspace(0, nq, nthreads + 1).astype(np.int64)
        L = lib()

        def run(k):
            lo, hi = int(cuts[k]), int(cuts[k + 1])
            if hi > lo:
                L.oracle_kdtree_query(self._h, q[lo:hi], hi - lo, idx[lo:hi], sq[lo:hi], found[lo:hi])
        with ThreadPoolExecutor(nthreads) as ex:
            list(ex.map(run, range(nthreads)))
        return idx, sq, found

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_kdtree_free(self._h)
            self._h = None


def search_nearest_neibor(xyz, nrm, idx, found):
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    nrm = np.ascontiguousarray(nrm, np.float32).reshape(-1, 3)
    nq = idx.shape[0]
    out6 = np.empty((nq, 6), np.float64)
    ok = np.empty(nq, np.uint8)
    lib().oracle_search_nearest_neibor(xyz, nrm, np.ascontiguousarray(idx, np.uint32),
                                       np.ascontiguousarray(found, np.uint8), nq, out6, ok)
    return out6, ok


GATE_MAPPER_LOCAL, GATE_MAPPER_GLOBAL, GATE_CONTROLLER = 0, 1, 2


def associate(X, out6, ok, max_range, gate_mode):
    X = np.ascontiguousarray(X, np.float64).reshape(-1, 3)
    nq = X.shape[0]
    out6 = np.ascontiguousarray(out6, np.float64)
    ok = np.ascontiguousarray(ok, np.uint8)
    abcd = np.empty((nq, 4), np.float64)
    typ = np.empty(nq, np.uint8)
    dist = np.empty(nq, np.float64)
    ang = np.empty(nq, np.float64)
    d2p = np.empty(nq, np.float64)
    mr = None
    if max_range is not None:
        mr_arr = np.ascontiguousarray(np.broadcast_to(np.asarray(max_range, np.float64), (nq,)))
        mr = mr_arr.ctypes.data_as(C.c_void_p)
    lib().oracle_associate(X, out6, ok, mr, nq, gate_mode, abcd, typ, dist, ang, d2p)
    return abcd, typ, dist, ang, d2p


def filter_lidar_outlier(X, lidar_xyz, typ, max_proj, max_icp):
    X = np.ascontiguousarray(X, np.float64).reshape(-1, 3)
    lx = np.ascontiguousarray(lidar_xyz, np.float64).reshape(-1, 3)
    typ = np.ascontiguousarray(typ, np.uint8)
    out = np.zeros(X.shape[0], np.uint8)
    L = lib()
    L.oracle_filter_lidar_outlier.restype = None
    L.oracle_filter_lidar_outlier.argtypes = [f64p, f64p, u8p, C.c_uint64, C.c_double, C.c_double, u8p]
    L.oracle_filter_lidar_outlier(X, lx, typ, X.shape[0], max_proj, max_icp, out)
    return out


def search_range_schedule(opt_num, kd_max=1.5, kd_min=0.2, drop=0.1):
    opt_num = np.ascontiguousarray(opt_num, np.int32)
    out = np.empty(opt_num.shape[0], np.float64)
    lib().oracle_search_range_schedule(opt_num, opt_num.shape[0], kd_max, kd_min, drop, out)
    return out


# ------------------------------------------------------------------ BA ----
CAMERA_MODELS = ["SIMPLE_PINHOLE", "PINHOLE", "SIMPLE_RADIAL", "RADIAL", "OPENCV", "OPENCV_FISHEYE",
                 "FULL_OPENCV", "FOV", "SIMPLE_RADIAL_FISHEYE", "RADIAL_FISHEYE", "THIN_PRISM_FISHEYE"]
NUM_PARAMS = [3, 4, 4, 5, 8, 8, 12, 5, 4, 5, 12]


def reproj_residual(model, q, t, X, cam, obs):
    r = np.empty(2, np.float64)
    a = [np.ascontiguousarray(v, np.float64) for v in (q, t, X, cam, obs)]
    lib().oracle_reproj_residual(model, *a, r)
    return r


def reproj_block(model, q, t, X, cam, obs):
    K = NUM_PARAMS[model]
    r = np.empty(2, np.float64)
    Jq = np.empty((2, 4)); Jt = np.empty((2, 3)); JX = np.empty((2, 3)); Jc = np.empty((2, K))
    a = [np.ascontiguousarray(v, np.float64) for v in (q, t, X, cam, obs)]
    lib().oracle_reproj_block(model, *a, r, Jq, Jt, JX, Jc)
    return r, Jq, Jt, JX, Jc


def lidar_block(X, abcd, w, strict=False):
    r = np.empty(1, np.float64)
    J = np.empty(3, np.float64)
    lib().oracle_lidar_block(np.ascontiguousarray(X, np.float64), np.ascontiguousarray(abcd, np.float64),
                             float(w), int(strict), r, J)
    return r[0], J


def loss(type_, scale, s):
    rho = np.empty(3, np.float64)
    lib().oracle_loss(type_, scale, s, rho)
    return rho


class BA:
    """Flat bundle-adjustment problem (same field meaning as include/pcdhip.h pcd_ba_desc)."""

    def __init__(self, cam_model, cam_params_list, poses, image_camera, points, obs_image, obs_point, obs_xy,
                 lidar_point=None, lidar_abcd=None, lidar_weight=None, image_const_pose=None,
                 image_const_tvec=None, point_const=None, loss_type=0, loss_scale=1.0):
        self.cam_model = np.ascontiguousarray(cam_model, np.int32)
        offs, flat = [], []
        for cp in cam_params_list:
            offs.append(len(flat))
            flat.extend(list(cp))
        self.cam_param_off = np.ascontiguousarray(offs, np.int32)
        self.cam_params = np.ascontiguousarray(flat, np.float64)
        self.poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
        self.image_camera = np.ascontiguousarray(image_camera, np.int32)
        self.points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
        self.obs_image = np.ascontiguousarray(obs_image, np.int32)
        self.obs_point = np.ascontiguousarray(obs_point, np.int32)
        self.obs_xy = np.ascontiguousarray(obs_xy, np.float64).reshape(-1, 2)
        nl = 0 if lidar_point is None else len(lidar_point)
        self.lidar_point = np.ascontiguousarray(lidar_point if nl else [], np.int32)
        self.lidar_abcd = np.ascontiguousarray(lidar_abcd if nl else [], np.float64).reshape(-1, 4)
        self.lidar_weight = np.ascontiguousarray(lidar_weight if nl else [], np.float64)
        I, P = self.poses.shape[0], self.points.shape[0]
        self.image_const_pose = np.ascontiguousarray(
            image_const_pose if image_const_pose is not None else np.zeros(I), np.uint8)
        self.image_const_tvec = np.ascontiguousarray(
            image_const_tvec if image_const_tvec is not None else np.zeros(I), np.uint8)
        self.point_const = np.ascontiguousarray(point_const if point_const is not None else np.zeros(P), np.uint8)
        self.loss_type, self.loss_scale = int(loss_type), float(loss_scale)
        self.stride = max(NUM_PARAMS[m] for m in self.cam_model)
        p = BAProblem()
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        p.num_cameras = len(self.cam_model); p.cam_model = vp(self.cam_model)
        p.cam_param_off = vp(self.cam_param_off); p.cam_params = vp(self.cam_params)
        p.num_images = I; p.poses = vp(self.poses); p.image_camera = vp(self.image_camera)
        p.image_const_pose = vp(self.image_const_pose); p.image_const_tvec = vp(self.image_const_tvec)
        p.num_points = P; p.points = vp(self.points); p.point_const = vp(self.point_const)
        p.num_obs = len(self.obs_image); p.obs_image = vp(self.obs_image); p.obs_point = vp(self.obs_point)
        p.obs_xy = vp(self.obs_xy)
        p.num_lidar = nl; p.lidar_point = vp(self.lidar_point); p.lidar_abcd = vp(self.lidar_abcd)
        p.lidar_weight = vp(self.lidar_weight)
        p.loss_type = self.loss_type; p.loss_scale = self.loss_scale; p.cam_jac_stride = self.stride
        self._p = p

    def evaluate_raw(self):
        O, Lc, S = len(self.obs_image), len(self.lidar_point), self.stride
        res = np.zeros(2 * O + Lc); Jq = np.zeros((O, 2, 4)); Jt = np.zeros((O, 2, 3)); JX = np.zeros((O, 2, 3))
        Jc = np.zeros((O, 2, S)); JL = np.zeros((Lc, 3))
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        lib().oracle_ba_evaluate_raw(C.byref(self._p), vp(res), vp(Jq), vp(Jt), vp(JX), vp(Jc), vp(JL))
        return res, Jq, Jt, JX, Jc, JL

    def residuals(self):
        """residual-only evaluation (plain-double functors): CostFunction::Evaluate(params, r, nullptr)"""
        res = np.zeros(2 * len(self.obs_image) + len(self.lidar_point))
        lib().oracle_ba_residuals(C.byref(self._p), res.ctypes.data_as(C.c_void_p))
        return res

    def _chunks(self, nthreads):
        """sub-problems over contiguous ranges of the observation / lidar blocks (pointer offsets, no copies)"""
        O, Lc = len(self.obs_image), len(self.lidar_point)
        oc = np.linspace(0, O, nthreads + 1).astype(np.int64)
        lc = np.linspace(0, Lc, nthreads + 1).astype(np.int64)
        out = []
        for k in range(nthreads):
            sp = BAProblem()
            C.pointer(sp)[0] = self._p
            o0, o1, l0, l1 = int(oc[k]), int(oc[k + 1]), int(lc[k]), int(lc[k + 1])
            base = lambda a, off: C.c_void_p(a.ctypes.data + off)
            sp.num_obs = o1 - o0
            sp.obs_image = base(self.obs_image, 4 * o0); sp.obs_point = base(self.obs_point, 4 * o0)
            sp.obs_xy = base(self.obs_xy, 16 * o0)
            sp.num_lidar = l1 - l0
            sp.lidar_point = base(self.lidar_point, 4 * l0); sp.lidar_abcd = base(self.lidar_abcd, 32 * l0)
            sp.lidar_weight = base(self.lidar_weight, 8 * l0)
            out.append((sp, o0, o1, l0, l1))
        return out

    def evaluate_mt(self, nthreads, jacobians=True):
        """Ceres-style threaded evaluation: residual blocks split over `nthreads` host threads, each evaluating
        its blocks with Jets (jacobians=True) or plain doubles.  bench.py's all-cores CPU baseline; outputs are
        per-thread scratch (Ceres writes into its own Jacobian storage) and are discarded."""
        from concurrent.futures import ThreadPoolExecutor
        S = self.stride
        L = lib()
        work = []
        for sp, o0, o1, l0, l1 in self._chunks(nthreads):
            n, m = o1 - o0, l1 - l0
            bufs = [np.empty(2 * n + m)]
            if jacobians:
                bufs += [np.empty((n, 2, 4)), np.empty((n, 2, 3)), np.empty((n, 2, 3)), np.empty((n, 2, S)),
                         np.empty((m, 3))]
            work.append((sp, bufs))

        def run(w):
            sp, bufs = w
            ptrs = [b.ctypes.data_as(C.c_void_p) for b in bufs]
            if jacobians:
                L.oracle_ba_evaluate_raw(C.byref(sp), *ptrs)
            else:
                L.oracle_ba_residuals(C.byref(sp), ptrs[0])
            return float(bufs[0] @ bufs[0])
        with ThreadPoolExecutor(nthreads) as ex:
            return 0.5 * sum(ex.map(run, work))

    def camera_blocks(self, refine, want_w=False):
        """refine: uint8 [len(cam_params)] (1 = optimised).  Returns H_cam [C][S][S], g_cam [C][S], E_cam [I][S][6],
        W_cam [O][S][3] or None -- with S = 12 (the C ABI's PCD_CAM_JAC_STRIDE)."""
        S = 12
        keep = self._p.cam_jac_stride
        self._p.cam_jac_stride = S
        C_, I, O = len(self.cam_model), self.poses.shape[0], len(self.obs_image)
        refine = np.ascontiguousarray(refine, np.uint8)
        assert refine.shape[0] == len(self.cam_params)
        H = np.zeros((C_, S, S)); g = np.zeros((C_, S)); E = np.zeros((I, S, 6))
        W = np.zeros((O, S, 3)) if want_w else None
        vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        L = lib()
        L.oracle_ba_camera_blocks.restype = None
        L.oracle_ba_camera_blocks.argtypes = [C.POINTER(BAProblem)] + [C.c_void_p] * 5
        L.oracle_ba_camera_blocks(C.byref(self._p), vp(refine), vp(H), vp(g), vp(E), vp(W))
        self._p.cam_jac_stride = keep
        return H, g, E, W

    def observation_errors(self):
        O = len(self.obs_image)
        sq = np.zeros(O)
        depth = np.zeros(O)
        L = lib()
        L.oracle_ba_observation_errors.restype = None
        L.oracle_ba_observation_errors.argtypes = [C.POINTER(BAProblem), C.c_void_p, C.c_void_p]
        L.oracle_ba_observation_errors(C.byref(self._p), sq.ctypes.data_as(C.c_void_p), depth.ctypes.data_as(C.c_void_p))
        return sq, depth

    def normal_equations(self, want_w=False):
        I, P, O = self.poses.shape[0], self.points.shape[0], len(self.obs_image)
        Himg = np.zeros((I, 6, 6)); gimg = np.zeros((I, 6)); Hpt = np.zeros((P, 3, 3)); gpt = np.zeros((P, 3))
        W = np.zeros((O, 6, 3)) if want_w else None
        vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        cost = lib().oracle_ba_normal_equations(C.byref(self._p), vp(Himg), vp(gimg), vp(Hpt), vp(gpt), vp(W))
        return cost, Himg, gimg, Hpt, gpt, W


def filter_tracks(sq_err, depth, obs_point, num_points, max_reproj_error):
    """Per-track reduce of the post-BA filters, restated from the reference (test infrastructure):
      Reconstruction::FilterPoints3DWithLargeReprojectionError  base/reconstruction.cc:1662-1712
      Reconstruction::FilterObservationsWithNegativeDepth       base/reconstruction.cc:837-855 (base/projection.cc:191-195)
      Reconstruction::ComputeMeanReprojectionError              base/reconstruction.cc:906-921
    sq_err / depth per observation (BA.observation_errors), obs_point [O] the point of every observation; a track =
    the observations of a point in ascending observation index.  Plain Python loops: small cases only."""
    O = len(obs_point)
    tracks = [[] for _ in range(num_points)]
    for o in range(O):
        tracks[int(obs_point[o])].append(o)
    max_sq = max_reproj_error * max_reproj_error
    obs_erase = np.zeros(O, np.uint8)
    point_delete = np.zeros(num_points, np.uint8)
    point_error = np.full(num_points, -1.0)
    num_filtered = 0
    for p, tr in enumerate(tracks):
        if len(tr) < 2:                                  # :1677-1681
            num_filtered += len(tr)
            point_delete[p] = 1
            obs_erase[tr] = 1
            continue
        to_delete, err_sum = [], 0.0
        for o in tr:                                     # :1687-1698
            if sq_err[o] > max_sq:
                to_delete.append(o)
            else:
                err_sum += np.sqrt(sq_err[o])
        if len(to_delete) >= len(tr) - 1:                # :1700-1702
            num_filtered += len(tr)
            point_delete[p] = 1
            obs_erase[tr] = 1
        else:                                            # :1703-1709 (DeleteObservation shortens the track first)
            num_filtered += len(to_delete)
            obs_erase[to_delete] = 1
            point_error[p] = err_sum / (len(tr) - len(to_delete))
    has = point_error >= 0
    mean = float(point_error[has].sum() / has.sum()) if has.any() else 0.0      # :906-921
    neg = (np.asarray(depth) < np.finfo(np.float64).eps).astype(np.uint8)       # !HasPointPositiveDepth
    return dict(obs_erase=obs_erase, obs_negative_depth=neg, point_delete=point_delete, point_error=point_error,
                num_filtered=num_filtered, mean_reproj_error=mean, num_points_with_error=int(has.sum()),
                num_negative_depth=int(neg.sum()))


# ------------------------------------------------- depth projection (N1) ----
class ProjOptions(C.Structure):
    """lidar/pcd_projection.h:31-47 (the numeric members)."""
    _fields_ = [("depth_image_scale", C.c_double), ("max_proj_scale", C.c_int32), ("min_proj_scale", C.c_int32),
                ("min_proj_dist", C.c_double), ("submap_length", C.c_float), ("submap_width", C.c_float),
                ("submap_height", C.c_float), ("choose_meter", C.c_float), ("min_lidar_proj_dist", C.c_double)]


class ProjImage(C.Structure):
    _fields_ = [("qvec", C.c_double * 4), ("tvec", C.c_double * 3), ("params", C.c_double * 8),
                ("width", C.c_uint64), ("height", C.c_uint64), ("feat_begin", C.c_uint64), ("feat_end", C.c_uint64)]


def proj_options(depth_image_scale=0.2, max_proj_scale=10, min_proj_scale=2, min_proj_dist=2.0, submap=1.0,
                 choose_meter=40.0, min_lidar_proj_dist=0.5):
    return ProjOptions(depth_image_scale, max_proj_scale, min_proj_scale, min_proj_dist, submap, submap, submap,
                       choose_meter, min_lidar_proj_dist)


def proj_scale_coeffs(opt, fx, fy):
    c4 = np.zeros(4, np.float64)
    L = lib()
    L.oracle_proj_scale_coeffs.restype = None
    L.oracle_proj_scale_coeffs.argtypes = [C.POINTER(ProjOptions), C.c_double, C.c_double, f64p]
    L.oracle_proj_scale_coeffs(C.byref(opt), fx, fy, c4)
    return c4


def proj_images(xyz, nrm, opt, coeffs, images, feat_xy):
    """images: list of dict(qvec, tvec, params[8], width, height, feat_begin, feat_end).
    Returns found, index, dist, lidar6, cam_xyz (SetNewImage #1 and #2 outputs), pairs."""
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    nrm = np.ascontiguousarray(nrm, np.float32).reshape(-1, 3)
    feat_xy = np.ascontiguousarray(feat_xy, np.float64).reshape(-1, 2)
    nf = feat_xy.shape[0]
    arr = (ProjImage * len(images))()
    for k, im in enumerate(images):
        arr[k] = ProjImage((C.c_double * 4)(*im["qvec"]), (C.c_double * 3)(*im["tvec"]),
                           (C.c_double * 8)(*im["params"]), im["width"], im["height"], im["feat_begin"],
                           im["feat_end"])
    found = np.zeros(nf, np.uint8)
    index = np.full(nf, 0xFFFFFFFF, np.uint32)      # rows that belong to no image stay "none"
    dist = np.zeros(nf, np.float32)
    L = lib()
    L.oracle_proj_images.restype = C.c_int64
    L.oracle_proj_images.argtypes = [f32p, C.c_uint64, C.POINTER(ProjOptions), f64p, C.c_uint64,
                                     C.POINTER(ProjImage), f64p, u8p, u32p, f32p]
    pairs = L.oracle_proj_images(xyz, xyz.shape[0], C.byref(opt), np.ascontiguousarray(coeffs, np.float64),
                                 len(images), arr, feat_xy, found, index, dist)
    assert pairs >= 0
    l6 = np.zeros((nf, 6), np.float64)
    L.oracle_proj_lidar6.restype = None
    L.oracle_proj_lidar6.argtypes = [f32p, f32p, C.c_uint64, u8p, u32p, f64p]
    L.oracle_proj_lidar6(xyz, nrm, nf, found, index, l6)
    cam = np.zeros((nf, 3), np.float64)
    L.oracle_proj_ray_plane.restype = None
    L.oracle_proj_ray_plane.argtypes = [f64p, C.c_uint64, f64p, u8p, f64p, f64p]
    for im in images:
        b, e = im["feat_begin"], im["feat_end"]
        if e > b:
            c = np.zeros((e - b, 3), np.float64)
            L.oracle_proj_ray_plane(np.ascontiguousarray(im["params"], np.float64), e - b,
                                    np.ascontiguousarray(feat_xy[b:e]), np.ascontiguousarray(found[b:e]),
                                    np.ascontiguousarray(l6[b:e]), c)
            cam[b:e] = c
    return found, index, dist, l6, cam, pairs
