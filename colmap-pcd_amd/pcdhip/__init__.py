"""pcdhip -- ctypes binding of libpcdhip.so (include/pcdhip.h).

Plumbing only: every computation happens in the HIP library.  There is no
CPU fallback; on a box without a gfx950 device `Cloud(...)`/`BA(...)` raise
PcdError(PCD_ERR_NO_DEVICE).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)            # colmap-pcd_amd/
LIB_PATH = os.environ.get("PCDHIP_LIB", os.path.join(_ROOT, "libpcdhip.so"))   # override: tuning variants only

PCD_OK, PCD_ERR_INVALID, PCD_ERR_NO_DEVICE, PCD_ERR_HIP, PCD_ERR_OOM, PCD_ERR_UNSUPPORTED = range(6)
NN_AUTO, NN_BRUTEFORCE, NN_FALLBACK_ONLY, NN_GRID = 0, 1, 2, 3
GATE_MAPPER_LOCAL, GATE_MAPPER_GLOBAL, GATE_CONTROLLER = 0, 1, 2
GATE_BOUNDED_SEARCH = 0x100   # OR-ed into a gate mode: search bounded by the gate (pcdhip.h)
LIDAR_NONE, LIDAR_ICP, LIDAR_ICP_GROUND = 0, 1, 2
LOSS_TRIVIAL, LOSS_SOFT_L1, LOSS_CAUCHY = 0, 1, 2
LAYOUT_XYZ_NRM, LAYOUT_AOS32 = 0, 1
KEY_NONE = 0x7FFFFFFFFFFFFFFF

CAMERA_MODELS = ["SIMPLE_PINHOLE", "PINHOLE", "SIMPLE_RADIAL", "RADIAL", "OPENCV", "OPENCV_FISHEYE",
                 "FULL_OPENCV", "FOV", "SIMPLE_RADIAL_FISHEYE", "RADIAL_FISHEYE", "THIN_PRISM_FISHEYE"]


class PcdError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"pcdhip status {status}: {msg}")
        self.status = status


class CloudOptions(C.Structure):
    _fields_ = [("device", C.c_int32), ("layout", C.c_int32), ("raw_lidar_frame", C.c_int32),
                ("cell_size", C.c_float), ("index_base", C.c_uint32), ("index_stride", C.c_uint32),
                ("reserved", C.c_int32 * 8)]


class CloudInfo(C.Structure):
    _fields_ = [("cell_size", C.c_float), ("origin", C.c_float * 3), ("dims", C.c_int32 * 3),
                ("block_dims", C.c_int32 * 3), ("num_indexed", C.c_uint64), ("occupied_cells", C.c_uint64),
                ("build_ms", C.c_double), ("bbox_lo", C.c_float * 3), ("bbox_hi", C.c_float * 3)]


class AssocOut(C.Structure):
    _fields_ = [("lidar_xyz", C.c_void_p), ("abcd", C.c_void_p), ("type", C.c_void_p), ("dist", C.c_void_p),
                ("angle", C.c_void_p), ("dist2plane", C.c_void_p), ("nn_idx", C.c_void_p),
                ("nn_sqdist", C.c_void_p)]


class AssocHit(C.Structure):
    _fields_ = [("lidar_xyz", C.c_double * 3), ("abcd", C.c_double * 4), ("dist", C.c_double), ("angle", C.c_double),
                ("query", C.c_uint32), ("type", C.c_uint8), ("pad", C.c_uint8 * 3)]


HIT_DTYPE = np.dtype([("lidar_xyz", np.float64, 3), ("abcd", np.float64, 4), ("dist", np.float64), ("angle", np.float64),
                      ("query", np.uint32), ("type", np.uint8), ("pad", np.uint8, 3)])
assert HIT_DTYPE.itemsize == 80 and C.sizeof(AssocHit) == 80


class BADesc(C.Structure):
    _fields_ = [("device", C.c_int32), ("num_cameras", C.c_int32), ("cam_model", C.c_void_p),
                ("cam_param_offset", C.c_void_p), ("cam_params", C.c_void_p), ("cam_params_len", C.c_uint64),
                ("num_images", C.c_int32), ("poses", C.c_void_p), ("image_camera", C.c_void_p),
                ("image_const_pose", C.c_void_p), ("image_const_tvec", C.c_void_p),
                ("num_points", C.c_int32), ("points", C.c_void_p), ("point_const", C.c_void_p),
                ("num_obs", C.c_uint64), ("obs_image", C.c_void_p), ("obs_point", C.c_void_p),
                ("obs_xy", C.c_void_p),
                ("num_lidar", C.c_uint64), ("lidar_point", C.c_void_p), ("lidar_abcd", C.c_void_p),
                ("lidar_weight", C.c_void_p),
                ("loss_type", C.c_int32), ("loss_scale", C.c_double), ("camera_refine", C.c_void_p),
                ("reserved", C.c_int32 * 8)]


CAM_JAC_STRIDE = 12


class BAOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("cost", "residuals", "jac_q", "jac_t", "jac_X", "jac_lidar",
                                           "H_img", "g_img", "H_pt", "g_pt", "W", "jac_cam",
                                           "H_cam", "g_cam", "E_cam", "W_cam")]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double)]


class NNStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("queries", "brick_groups", "staged_points", "fallback_queries",
                                           "fallback_points", "pair_evals")]


# every symbol include/pcdhip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "pcd_last_error", "pcd_version", "pcd_device_count",
    "pcd_cloud_options_default", "pcd_cloud_create", "pcd_cloud_destroy", "pcd_cloud_size",
    "pcd_cloud_get_info", "pcd_cloud_download",
    "pcd_nn_query", "pcd_nn_query_algo", "pcd_nn_query_device", "pcd_nn_refine_device",
    "pcd_associate", "pcd_associate_device", "pcd_assoc_staging", "pcd_associate_staged", "pcd_nn_winner_payload_device",
    "pcd_associate_from_payload_device", "pcd_search_range_schedule",
    "pcd_camera_num_params", "pcd_camera_param_groups", "pcd_ba_create", "pcd_ba_destroy", "pcd_ba_set_parameters", "pcd_ba_set_camera_parameters",
    "pcd_ba_evaluate", "pcd_ba_evaluate_device", "pcd_ba_device_parameters",
    "pcd_profile_enable", "pcd_profile_only", "pcd_profile_reset", "pcd_profile_get", "pcd_nn_last_stats",
    "pcd_sift_match", "pcd_sift_match_device", "pcd_sift_match_batch", "pcd_sift_match_batch_device",
    "pcd_filter_lidar_outlier_device", "pcd_ba_observation_errors", "pcd_ba_observation_errors_device",
    "pcd_proj_default_options", "pcd_proj_create", "pcd_proj_destroy", "pcd_proj_num_submaps",
    "pcd_proj_last_pairs", "pcd_proj_scale_coeffs", "pcd_proj_set_new_images",
    "pcd_sift_matcher_create", "pcd_sift_matcher_destroy", "pcd_sift_matcher_set_max_sift",
    "pcd_sift_matcher_set_descriptors", "pcd_sift_matcher_match",
    "pcd_ba_evaluate_blocks", "pcd_ba_filter_tracks", "pcd_ba_filter_tracks_device",
    "pcd_cloud_create_sharded", "pcd_cloud_shards_destroy", "pcd_cloud_shards_count", "pcd_cloud_shards_size",
    "pcd_cloud_shards_get", "pcd_nn_query_sharded", "pcd_associate_sharded",
]


class ProjOptions(C.Structure):
    """lidar/pcd_projection.h:31-47 (numeric members)."""
    _fields_ = [("depth_image_scale", C.c_double), ("max_proj_scale", C.c_int32), ("min_proj_scale", C.c_int32),
                ("min_proj_dist", C.c_double), ("submap_length", C.c_float), ("submap_width", C.c_float),
                ("submap_height", C.c_float), ("choose_meter", C.c_float), ("min_lidar_proj_dist", C.c_double)]


class ProjImage(C.Structure):
    _fields_ = [("qvec", C.c_double * 4), ("tvec", C.c_double * 3), ("params", C.c_double * 8),
                ("width", C.c_uint64), ("height", C.c_uint64), ("feat_begin", C.c_uint64), ("feat_end", C.c_uint64)]

_LIB = None


def build():
    """Compile libpcdhip.so in-tree with hipcc for gfx950 (works without a GPU)."""
    subprocess.check_call(["make", "-s", "-j4", "-C", _ROOT, "libpcdhip.so"])


def lib():
    """Load the HIP library.  Fails loudly if it is missing: there is no other backend."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise PcdError(PCD_ERR_NO_DEVICE, f"{LIB_PATH} not built; run __graft_entry__.build() or make -C colmap-pcd_amd")
    # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64.so.7 / libhsa-runtime64.
    # If libpcdhip pulled in /opt/rocm's copy first, torch would later mix the two and find no GPU, so
    # when torch is installed let it load its runtime first; libpcdhip then binds to the same soname.
    if os.environ.get("PCDHIP_NO_TORCH_PRELOAD", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    L.pcd_last_error.restype = C.c_char_p
    L.pcd_cloud_size.restype = C.c_uint64
    L.pcd_cloud_size.argtypes = [C.c_void_p]
    L.pcd_cloud_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(CloudOptions), C.POINTER(C.c_void_p)]
    L.pcd_cloud_destroy.argtypes = [C.c_void_p]
    L.pcd_cloud_destroy.restype = None
    L.pcd_cloud_get_info.argtypes = [C.c_void_p, C.POINTER(CloudInfo)]
    L.pcd_cloud_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.pcd_nn_query_algo.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pcd_nn_query.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pcd_nn_query_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
    L.pcd_associate.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int,
                                C.POINTER(AssocOut)]
    L.pcd_associate_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int,
                                       C.c_void_p, C.POINTER(AssocOut), C.c_void_p]
    L.pcd_nn_winner_payload_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.pcd_associate_from_payload_device.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                                    C.c_int, C.c_void_p, C.c_void_p, C.POINTER(AssocOut),
                                                    C.c_void_p]
    L.pcd_search_range_schedule.argtypes = [C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_void_p]
    L.pcd_profile_get.argtypes = [C.POINTER(KernelTime), C.c_int, C.POINTER(C.c_int)]
    L.pcd_nn_last_stats.argtypes = [C.c_void_p, C.POINTER(NNStats)]
    if hasattr(L, "pcd_ba_create"):
        L.pcd_ba_create.argtypes = [C.POINTER(BADesc), C.POINTER(C.c_void_p)]
        L.pcd_ba_destroy.argtypes = [C.c_void_p]
        L.pcd_ba_destroy.restype = None
        L.pcd_ba_set_parameters.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.pcd_ba_evaluate.argtypes = [C.c_void_p, C.POINTER(BAOut)]
        L.pcd_ba_evaluate_device.argtypes = [C.c_void_p, C.POINTER(BAOut), C.c_void_p]
        L.pcd_ba_device_parameters.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    _LIB = L
    return L


def _check(st):
    if st != PCD_OK:
        raise PcdError(st, lib().pcd_last_error().decode(errors="replace"))


def device_count():
    return lib().pcd_device_count()


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _ptr(x):
    """numpy array -> host pointer; int -> raw (device) pointer; torch tensor -> data_ptr()."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        return x.ctypes.data_as(C.c_void_p)
    if isinstance(x, int):
        return C.c_void_p(x)
    return C.c_void_p(x.data_ptr())


class Cloud:
    """Device-resident LiDAR cloud index (reference: lidar::PointCloudProcess + lidar::Kdtree)."""

    def __init__(self, xyz, nrm=None, device=0, raw_lidar_frame=True, cell_size=0.0, layout=LAYOUT_XYZ_NRM,
                 index_base=0, index_stride=1):
        xyz = np.ascontiguousarray(xyz, np.float32)
        n = xyz.shape[0] if xyz.ndim > 1 else 0
        if layout == LAYOUT_XYZ_NRM:
            xyz = xyz.reshape(-1, 3)
            nrm = np.ascontiguousarray(nrm, np.float32).reshape(-1, 3)
            assert nrm.shape[0] == xyz.shape[0]
            n = xyz.shape[0]
        else:
            xyz = xyz.reshape(-1, 8)
            n = xyz.shape[0]
        o = CloudOptions()
        lib().pcd_cloud_options_default(C.byref(o))
        o.device, o.layout, o.raw_lidar_frame, o.cell_size = device, layout, int(raw_lidar_frame), cell_size
        o.index_base, o.index_stride = index_base, index_stride
        h = C.c_void_p()
        self._h = None
        _check(lib().pcd_cloud_create(_vp(xyz), _vp(nrm), n, C.byref(o), C.byref(h)))
        self._h = h
        self.device = device

    def close(self):
        if self._h:
            lib().pcd_cloud_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: module globals may already be gone
            pass

    def __len__(self):
        return int(lib().pcd_cloud_size(self._h))

    def info(self):
        i = CloudInfo()
        _check(lib().pcd_cloud_get_info(self._h, C.byref(i)))
        return dict(cell_size=i.cell_size, origin=list(i.origin), dims=list(i.dims), block_dims=list(i.block_dims),
                    num_indexed=i.num_indexed, occupied_cells=i.occupied_cells, build_ms=i.build_ms,
                    bbox_lo=list(i.bbox_lo), bbox_hi=list(i.bbox_hi))

    def download(self):
        n = len(self)
        xyz = np.empty((n, 3), np.float32)
        nrm = np.empty((n, 3), np.float32)
        _check(lib().pcd_cloud_download(self._h, _vp(xyz), _vp(nrm)))
        return xyz, nrm

    def nn(self, q, algo=NN_AUTO):
        """Kdtree::GetClosestPoint for a batch: returns (idx uint32, sqdist float32, found uint8)."""
        q = np.ascontiguousarray(q, np.float64).reshape(-1, 3)
        Q = q.shape[0]
        idx = np.empty(Q, np.uint32)
        sq = np.empty(Q, np.float32)
        found = np.empty(Q, np.uint8)
        _check(lib().pcd_nn_query_algo(self._h, _vp(q), Q, algo, _vp(idx), _vp(sq), _vp(found)))
        return idx, sq, found

    def nn_device(self, d_q, Q, d_keys, algo=NN_AUTO, stream=0):
        _check(lib().pcd_nn_query_device(self._h, _ptr(d_q), Q, algo, _ptr(d_keys), C.c_void_p(stream)))

    def nn_refine_device(self, d_q, Q, d_keys, d_skip=None, stream=0):
        """second phase of a sharded search: keys in/out (pcd_nn_refine_device)"""
        L = lib()
        L.pcd_nn_refine_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        _check(L.pcd_nn_refine_device(self._h, _ptr(d_q), Q, _ptr(d_skip), _ptr(d_keys), C.c_void_p(stream)))

    def associate(self, q, max_range=None, gate_mode=GATE_MAPPER_LOCAL):
        q = np.ascontiguousarray(q, np.float64).reshape(-1, 3)
        Q = q.shape[0]
        out = dict(lidar_xyz=np.empty((Q, 3)), abcd=np.empty((Q, 4)), type=np.empty(Q, np.uint8),
                   dist=np.empty(Q), angle=np.empty(Q), dist2plane=np.empty(Q),
                   nn_idx=np.empty(Q, np.uint32), nn_sqdist=np.empty(Q, np.float32))
        ao = AssocOut(*[_vp(out[k]) for k in ("lidar_xyz", "abcd", "type", "dist", "angle", "dist2plane",
                                               "nn_idx", "nn_sqdist")])
        mr, mrc = None, 0
        if (gate_mode & 0xFF) != GATE_CONTROLLER:
            mr = np.ascontiguousarray(np.atleast_1d(np.asarray(max_range, np.float64)))
            mrc = mr.shape[0]
        _check(lib().pcd_associate(self._h, _vp(q), Q, _vp(mr), mrc, gate_mode, C.byref(ao)))
        return out

    def staging(self, Q):
        """pinned input buffers of the staged host path: (q_xyz [Q][3], max_range [Q]) numpy views"""
        L = lib()
        L.pcd_assoc_staging.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        a, b = C.c_void_p(), C.c_void_p()
        _check(L.pcd_assoc_staging(self._h, Q, C.byref(a), C.byref(b)))
        q = np.ctypeslib.as_array(C.cast(a, C.POINTER(C.c_double)), shape=(max(Q, 1), 3))[:Q]
        mr = np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_double)), shape=(max(Q, 1),))[:Q]
        return q, mr

    def associate_staged(self, Q, mr_count, gate_mode=GATE_MAPPER_LOCAL):
        """runs on the buffers of staging(); returns the accepted associations as a structured numpy view (80-byte
        records, ascending query) of the handle's pinned result buffer -- valid until the next call"""
        L = lib()
        L.pcd_associate_staged.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_void_p),
                                           C.POINTER(C.c_uint64)]
        h, n = C.c_void_p(), C.c_uint64(0)
        _check(L.pcd_associate_staged(self._h, Q, mr_count, gate_mode, C.byref(h), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, HIT_DTYPE)
        buf = (C.c_char * (80 * n.value)).from_address(h.value)
        return np.frombuffer(buf, dtype=HIT_DTYPE, count=n.value)

    def associate_device(self, d_q, Q, d_max_range, mr_count, gate_mode, d_out, d_keys_in=None, stream=0):
        ao = AssocOut(*[_ptr(d_out.get(k)) for k in ("lidar_xyz", "abcd", "type", "dist", "angle", "dist2plane",
                                                      "nn_idx", "nn_sqdist")])
        _check(lib().pcd_associate_device(self._h, _ptr(d_q), Q, _ptr(d_max_range), mr_count, gate_mode,
                                          _ptr(d_keys_in), C.byref(ao), C.c_void_p(stream)))

    def winner_payload_device(self, d_keys, Q, d_payload, stream=0):
        _check(lib().pcd_nn_winner_payload_device(self._h, _ptr(d_keys), Q, _ptr(d_payload), C.c_void_p(stream)))

    def last_stats(self):
        s = NNStats()
        _check(lib().pcd_nn_last_stats(self._h, C.byref(s)))
        return {n: getattr(s, n) for n, _ in NNStats._fields_}


def associate_from_payload_device(device, d_q, Q, d_max_range, mr_count, gate_mode, d_keys, d_payload, d_out,
                                  stream=0):
    ao = AssocOut(*[_ptr(d_out.get(k)) for k in ("lidar_xyz", "abcd", "type", "dist", "angle", "dist2plane",
                                                  "nn_idx", "nn_sqdist")])
    _check(lib().pcd_associate_from_payload_device(device, _ptr(d_q), Q, _ptr(d_max_range), mr_count, gate_mode,
                                                   _ptr(d_keys), _ptr(d_payload), C.byref(ao), C.c_void_p(stream)))


class ShardReduce(C.Structure):
    """pcd_shard_reduce: the exchange steps of the sharded search (NULL members: the library's own peer-copy reduction)"""
    MINFN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_uint64)
    SUMFN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_uint64)
    _fields_ = [("min_u64", MINFN), ("sum_i32", SUMFN), ("user", C.c_void_p)]


class ShardedCloud:
    """One cloud over several devices of this process (pcd_cloud_create_sharded): what a multi-GPU C++ host builds in
    LoadPointcloud.  devices may repeat (tests put every shard on device 0)."""

    def __init__(self, xyz, nrm, devices, raw_lidar_frame=True, cell_size=0.0):
        L = lib()
        xyz = np.ascontiguousarray(xyz, np.float32)
        nrm = np.ascontiguousarray(nrm, np.float32)
        o = CloudOptions()
        L.pcd_cloud_options_default(C.byref(o))
        o.raw_lidar_frame = int(raw_lidar_frame)
        o.cell_size = cell_size
        dv = (C.c_int * len(devices))(*devices)
        L.pcd_cloud_create_sharded.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(CloudOptions), C.POINTER(C.c_int),
                                               C.c_int, C.POINTER(C.c_void_p)]
        L.pcd_cloud_shards_destroy.argtypes = [C.c_void_p]
        L.pcd_cloud_shards_destroy.restype = None
        L.pcd_cloud_shards_size.argtypes = [C.c_void_p]
        L.pcd_cloud_shards_size.restype = C.c_uint64
        self._h = C.c_void_p()
        n = xyz.shape[0]
        _check(L.pcd_cloud_create_sharded(_vp(xyz) if n else None, _vp(nrm) if n else None, n, C.byref(o), dv, len(devices),
                                          C.byref(self._h)))

    def __len__(self):
        return int(lib().pcd_cloud_shards_size(self._h))

    def close(self):
        if self._h:
            lib().pcd_cloud_shards_destroy(self._h)
            self._h = None

    def nn(self, q, reduce=None):
        q = np.ascontiguousarray(q, np.float64).reshape(-1, 3)
        Q = q.shape[0]
        idx = np.empty(Q, np.uint32); sq = np.empty(Q, np.float32); found = np.empty(Q, np.uint8)
        L = lib()
        L.pcd_nn_query_sharded.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _check(L.pcd_nn_query_sharded(self._h, _vp(q) if Q else None, Q, C.byref(reduce) if reduce is not None else None,
                                      _vp(idx), _vp(sq), _vp(found)))
        return idx, sq, found

    def associate(self, q, max_range, gate_mode=0, reduce=None):
        q = np.ascontiguousarray(q, np.float64).reshape(-1, 3)
        Q = q.shape[0]
        mr = np.ascontiguousarray(np.atleast_1d(max_range), np.float64)
        out = dict(lidar_xyz=np.zeros((Q, 3)), abcd=np.zeros((Q, 4)), type=np.zeros(Q, np.uint8), dist=np.zeros(Q),
                   angle=np.zeros(Q), dist2plane=np.zeros(Q), nn_idx=np.zeros(Q, np.uint32), nn_sqdist=np.zeros(Q, np.float32))
        ao = AssocOut(*[_vp(out[k]) for k in ("lidar_xyz", "abcd", "type", "dist", "angle", "dist2plane", "nn_idx", "nn_sqdist")])
        L = lib()
        L.pcd_associate_sharded.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p,
                                            C.POINTER(AssocOut)]
        _check(L.pcd_associate_sharded(self._h, _vp(q), Q, _vp(mr), mr.shape[0], gate_mode,
                                       C.byref(reduce) if reduce is not None else None, C.byref(ao)))
        return out


def sift_match(d1, d2, max_ratio=0.8, max_distance=0.7, cross_check=True, device=0):
    """MatchSiftFeaturesCPUBruteForce semantics on the GPU: returns matches [M][2] uint32"""
    d1 = np.ascontiguousarray(d1, np.uint8).reshape(-1, 128)
    d2 = np.ascontiguousarray(d2, np.uint8).reshape(-1, 128)
    n1, n2 = d1.shape[0], d2.shape[0]
    m = np.zeros((max(n1, 1), 2), np.uint32)
    cnt = C.c_int32(0)
    L = lib()
    L.pcd_sift_match.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int,
                                 C.c_void_p, C.POINTER(C.c_int32)]
    _check(L.pcd_sift_match(device, _vp(d1) if n1 else None, n1, _vp(d2) if n2 else None, n2, max_ratio,
                            max_distance, int(cross_check), _vp(m), C.byref(cnt)))
    return m[:cnt.value].copy()


def sift_match_device(d_d1, n1, d_d2, n2, d_m12, d_m21, d_matches, d_count, max_ratio=0.8, max_distance=0.7,
                      cross_check=True, device=0, stream=0):
    L = lib()
    L.pcd_sift_match_device.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                        C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _check(L.pcd_sift_match_device(device, _ptr(d_d1), n1, _ptr(d_d2), n2, max_ratio, max_distance, int(cross_check),
                                   _ptr(d_m12), _ptr(d_m21), _ptr(d_matches), _ptr(d_count), C.c_void_p(stream)))


def _sift_arena(descriptors):
    """list of [n_i][128] uint8 arrays -> (arena [sum n_i][128], first_row [len + 1] uint64)"""
    ds = [np.ascontiguousarray(d, np.uint8).reshape(-1, 128) for d in descriptors]
    first = np.zeros(len(ds) + 1, np.uint64)
    first[1:] = np.cumsum([d.shape[0] for d in ds])
    arena = np.concatenate(ds, axis=0) if ds and first[-1] else np.zeros((0, 128), np.uint8)
    return np.ascontiguousarray(arena), first


def sift_match_batch(descriptors, pairs, max_ratio=0.8, max_distance=0.7, cross_check=True, device=0):
    """SiftFeatureMatcher::Match(image_pairs) (feature/matching.cc:798): `descriptors` = one [n_i][128] uint8 array
    per image, `pairs` = [P][2] image indices.  Returns a list of P match arrays [M_p][2] uint32, each equal to
    sift_match(descriptors[a], descriptors[b])."""
    arena, first = _sift_arena(descriptors)
    pairs = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
    P = pairs.shape[0]
    off = np.zeros(P + 1, np.uint64)
    n1 = (first[1:] - first[:-1])[pairs[:, 0]] if P else np.zeros(0, np.uint64)
    cap = int(n1.sum())
    m = np.zeros((max(cap, 1), 2), np.uint32)
    L = lib()
    L.pcd_sift_match_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float,
                                       C.c_float, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]
    _check(L.pcd_sift_match_batch(device, _vp(arena) if arena.shape[0] else None, _vp(first), len(descriptors),
                                  _vp(pairs) if P else None, P, max_ratio, max_distance, int(cross_check), _vp(m), cap,
                                  _vp(off)))
    return [m[int(off[p]):int(off[p + 1])].copy() for p in range(P)]


def exhaustive_blocks(n_images, block_size=50):
    """The image-pair lists of ExhaustiveFeatureMatcher::Run (feature/matching.cc:902-960), one per block pair, in the
    reference's order: blocks of `block_size` consecutive images, pair (i1, i2) taken when
    (i1 > i2 and i1 % B <= i2 % B) or (i1 < i2 and i1 % B < i2 % B) -- every unordered pair exactly once.  Each list
    is what the reference hands to SiftFeatureMatcher::Match(image_pairs) = sift_match_batch[_device]."""
    B = int(block_size)
    for s1 in range(0, n_images, B):
        e1 = min(n_images, s1 + B)
        for s2 in range(0, n_images, B):
            e2 = min(n_images, s2 + B)
            i1, i2 = np.meshgrid(np.arange(s1, e1), np.arange(s2, e2), indexing="ij")
            b1, b2 = i1 % B, i2 % B
            keep = ((i1 > i2) & (b1 <= b2)) | ((i1 < i2) & (b1 < b2))
            yield np.stack([i1[keep], i2[keep]], axis=1).astype(np.uint32)


def sift_match_batch_device(d_arena, first_row, pairs, d_matches, match_offset, d_counts, max_ratio=0.8,
                            max_distance=0.7, cross_check=True, device=0, stream=0):
    """device form: first_row / pairs / match_offset are numpy (host) arrays, the rest torch device tensors"""
    first_row = np.ascontiguousarray(first_row, np.uint64)
    pairs = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
    match_offset = np.ascontiguousarray(match_offset, np.uint64)
    L = lib()
    L.pcd_sift_match_batch_device.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float,
                                              C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _check(L.pcd_sift_match_batch_device(device, _ptr(d_arena), _vp(first_row), first_row.shape[0] - 1, _vp(pairs),
                                         pairs.shape[0], max_ratio, max_distance, int(cross_check), _ptr(d_matches),
                                         _vp(match_offset), _ptr(d_counts), C.c_void_p(stream)))


def filter_lidar_outlier_device(d_points, d_lidar_xyz, d_type, n, max_proj, max_icp, d_erase, device=0, stream=0):
    L = lib()
    L.pcd_filter_lidar_outlier_device.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                                  C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    _check(L.pcd_filter_lidar_outlier_device(device, _ptr(d_points), _ptr(d_lidar_xyz), _ptr(d_type), n, max_proj,
                                             max_icp, _ptr(d_erase), C.c_void_p(stream)))


class Projector:
    """Depth-projection association over a Cloud (reference: lidar::PcdProj, lidar/pcd_projection.h:48-185)."""

    def __init__(self, cloud, options=None, **kw):
        L = lib()
        L.pcd_proj_create.argtypes = [C.c_void_p, C.POINTER(ProjOptions), C.POINTER(C.c_void_p)]
        L.pcd_proj_destroy.argtypes = [C.c_void_p]
        L.pcd_proj_destroy.restype = None
        L.pcd_proj_num_submaps.argtypes = [C.c_void_p]
        L.pcd_proj_num_submaps.restype = C.c_uint64
        L.pcd_proj_last_pairs.argtypes = [C.c_void_p]
        L.pcd_proj_last_pairs.restype = C.c_uint64
        L.pcd_proj_scale_coeffs.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        L.pcd_proj_set_new_images.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(ProjImage), C.c_uint64] + \
            [C.c_void_p] * 6
        if options is None:
            options = ProjOptions()
            L.pcd_proj_default_options(C.byref(options))
        for k, v in kw.items():
            setattr(options, k, v)
        self.options = options
        self._cloud = cloud          # keep the cloud alive
        self._h = None
        h = C.c_void_p()
        _check(L.pcd_proj_create(cloud._h, C.byref(options), C.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            lib().pcd_proj_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_submaps(self):
        return int(lib().pcd_proj_num_submaps(self._h))

    @property
    def last_pairs(self):
        return int(lib().pcd_proj_last_pairs(self._h))

    def scale_coeffs(self, set_to=None):
        c4 = np.zeros(4, np.float64) if set_to is None else np.ascontiguousarray(set_to, np.float64)
        latched = C.c_int(0)
        _check(lib().pcd_proj_scale_coeffs(self._h, 0 if set_to is None else 1, _vp(c4), C.byref(latched)))
        return c4, bool(latched.value)

    def set_new_images(self, images, feat_xy):
        """images: list of dict(qvec, tvec, params[8], width, height, feat_begin, feat_end).
        Returns found, lidar_index, dist, lidar6, cam_xyz."""
        feat_xy = np.ascontiguousarray(feat_xy, np.float64).reshape(-1, 2)
        nf = feat_xy.shape[0]
        arr = (ProjImage * max(len(images), 1))()
        for k, im in enumerate(images):
            arr[k] = ProjImage((C.c_double * 4)(*im["qvec"]), (C.c_double * 3)(*im["tvec"]),
                               (C.c_double * 8)(*im["params"]), im["width"], im["height"], im["feat_begin"],
                               im["feat_end"])
        found = np.zeros(nf, np.uint8)
        index = np.zeros(nf, np.uint32)
        dist = np.zeros(nf, np.float32)
        l6 = np.zeros((nf, 6), np.float64)
        cam = np.zeros((nf, 3), np.float64)
        _check(lib().pcd_proj_set_new_images(self._h, len(images), arr, nf, _vp(feat_xy), _vp(found), _vp(index),
                                             _vp(dist), _vp(l6), _vp(cam)))
        return found, index, dist, l6, cam


def camera_param_groups(model_id):
    """0 focal length / 1 principal point / 2 extra parameter, per parameter of the model"""
    k = lib().pcd_camera_num_params(int(model_id))
    g = np.zeros(max(k, 1), np.uint8)
    L = lib()
    L.pcd_camera_param_groups.argtypes = [C.c_int, C.c_void_p]
    _check(L.pcd_camera_param_groups(int(model_id), _vp(g)))
    return g[:k]


def camera_refine_mask(cam_model, refine_focal_length, refine_principal_point, refine_extra_params,
                       constant_cameras=()):
    """BundleAdjuster::ParameterizeCameras (optim/bundle_adjustment.cc:1047-1100) as a flat mask over cam_params"""
    sel = [bool(refine_focal_length), bool(refine_principal_point), bool(refine_extra_params)]
    out = []
    for c, m in enumerate(cam_model):
        g = camera_param_groups(m)
        out.extend([0] * len(g) if c in constant_cameras else [int(sel[x]) for x in g])
    return np.array(out, np.uint8)


def search_range_schedule(opt_num, kd_max=1.5, kd_min=0.2, drop=0.1):
    opt_num = np.ascontiguousarray(opt_num, np.int32)
    out = np.empty(opt_num.shape[0], np.float64)
    _check(lib().pcd_search_range_schedule(_vp(opt_num), opt_num.shape[0], kd_max, kd_min, drop, _vp(out)))
    return out


def set_nn_tuning(brick_cells=0, halo_cells=-1, collect_stats=0):
    lib().pcd_nn_set_tuning(int(brick_cells), int(halo_cells), int(collect_stats))


def set_nn_search(kernel=0):
    """first stage of the grid path: 0 = clipped brick kernel (default), 1 = the same with the clip off, 2 = round 3's kernel"""
    _check(lib().pcd_nn_set_search(int(kernel)))


def set_sift_tuning(nchunk=0, batch_partials=0):
    """tests / fuzzing: column chunks per stripe walk and bytes of partial results per sub-batch (0 = library's choice)"""
    L = lib()
    L.pcd_sift_set_tuning.argtypes = [C.c_int, C.c_uint64]
    _check(L.pcd_sift_set_tuning(int(nchunk), int(batch_partials)))


def set_nn_bookkeeping(radix_sort=0):
    _check(lib().pcd_nn_set_bookkeeping(int(radix_sort)))


def profile_enable(on=True):
    lib().pcd_profile_enable(int(on))


def profile_only(scope=None):
    """time just one scope (None: all of them)"""
    L = lib()
    L.pcd_profile_only.argtypes = [C.c_char_p]
    L.pcd_profile_only(scope.encode() if scope else None)


def profile_reset():
    lib().pcd_profile_reset()


def profile_get():
    arr = (KernelTime * 64)()
    n = C.c_int(0)
    lib().pcd_profile_get(arr, 64, C.byref(n))
    return {arr[i].name.decode(): (int(arr[i].launches), float(arr[i].total_ms)) for i in range(min(n.value, 64))}


class BA:
    """Flat BA problem on the device (reference: what BundleAdjuster::SetUp*ByLidar builds for Ceres)."""

    def __init__(self, cam_model, cam_params_list, poses, image_camera, points, obs_image, obs_point, obs_xy,
                 lidar_point=None, lidar_abcd=None, lidar_weight=None, image_const_pose=None,
                 image_const_tvec=None, point_const=None, loss_type=LOSS_TRIVIAL, loss_scale=1.0, device=0,
                 camera_refine=None):
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
        self.I, self.P, self.O, self.L = self.poses.shape[0], self.points.shape[0], len(self.obs_image), nl
        self.image_const_pose = None if image_const_pose is None else np.ascontiguousarray(image_const_pose, np.uint8)
        self.image_const_tvec = None if image_const_tvec is None else np.ascontiguousarray(image_const_tvec, np.uint8)
        self.point_const = None if point_const is None else np.ascontiguousarray(point_const, np.uint8)
        d = BADesc()
        d.device = device
        d.num_cameras = len(self.cam_model); d.cam_model = _vp(self.cam_model)
        d.cam_param_offset = _vp(self.cam_param_off); d.cam_params = _vp(self.cam_params)
        d.cam_params_len = len(self.cam_params)
        d.num_images = self.I; d.poses = _vp(self.poses); d.image_camera = _vp(self.image_camera)
        d.image_const_pose = _vp(self.image_const_pose); d.image_const_tvec = _vp(self.image_const_tvec)
        d.num_points = self.P; d.points = _vp(self.points); d.point_const = _vp(self.point_const)
        d.num_obs = self.O; d.obs_image = _vp(self.obs_image); d.obs_point = _vp(self.obs_point)
        d.obs_xy = _vp(self.obs_xy)
        d.num_lidar = nl; d.lidar_point = _vp(self.lidar_point); d.lidar_abcd = _vp(self.lidar_abcd)
        d.lidar_weight = _vp(self.lidar_weight)
        d.loss_type, d.loss_scale = int(loss_type), float(loss_scale)
        self.camera_refine = None if camera_refine is None else np.ascontiguousarray(camera_refine, np.uint8)
        if self.camera_refine is not None:
            assert self.camera_refine.shape[0] == len(self.cam_params)
            d.camera_refine = _vp(self.camera_refine)
        self.C = len(self.cam_model)
        h = C.c_void_p()
        self._h = None
        _check(lib().pcd_ba_create(C.byref(d), C.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            lib().pcd_ba_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: module globals may already be gone
            pass

    def set_parameters(self, poses=None, points=None):
        p = None if poses is None else np.ascontiguousarray(poses, np.float64)
        x = None if points is None else np.ascontiguousarray(points, np.float64)
        _check(lib().pcd_ba_set_parameters(self._h, _vp(p), _vp(x)))

    def evaluate(self, want=("cost", "residuals", "jac_q", "jac_t", "jac_X", "jac_lidar", "H_img", "g_img", "H_pt",
                             "g_pt")):
        shapes = dict(cost=(1,), residuals=(2 * self.O + self.L,), jac_q=(self.O, 2, 4), jac_t=(self.O, 2, 3),
                      jac_X=(self.O, 2, 3), jac_lidar=(self.L, 3), H_img=(self.I, 6, 6), g_img=(self.I, 6),
                      H_pt=(self.P, 3, 3), g_pt=(self.P, 3), W=(self.O, 6, 3), jac_cam=(self.O, 2, CAM_JAC_STRIDE),
                      H_cam=(self.C, CAM_JAC_STRIDE, CAM_JAC_STRIDE), g_cam=(self.C, CAM_JAC_STRIDE),
                      E_cam=(self.I, CAM_JAC_STRIDE, 6), W_cam=(self.O, CAM_JAC_STRIDE, 3))
        out = {k: np.zeros(shapes[k]) for k in want}
        bo = BAOut(*[_vp(out.get(n)) for n, _ in BAOut._fields_])
        _check(lib().pcd_ba_evaluate(self._h, C.byref(bo)))
        return out

    def evaluate_device(self, d_out, stream=0):
        bo = BAOut(*[_ptr(d_out.get(n)) for n, _ in BAOut._fields_])
        _check(lib().pcd_ba_evaluate_device(self._h, C.byref(bo), C.c_void_p(stream)))

    def observation_errors(self):
        """(squared reprojection error, camera-frame depth) per observation: inputs of the post-BA filters"""
        sq = np.empty(self.O)
        depth = np.empty(self.O)
        L = lib()
        L.pcd_ba_observation_errors.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _check(L.pcd_ba_observation_errors(self._h, _vp(sq), _vp(depth)))
        return sq, depth

    def filter_tracks(self, max_reproj_error):
        """post-BA filters reduced per track on the device (pcd_ba_filter_tracks): dict of obs_erase [O],
        obs_negative_depth [O], point_delete [P], point_error [P] (-1: no error), num_filtered, mean_reproj_error,
        num_points_with_error, num_negative_depth"""
        class FO(C.Structure):
            _fields_ = [(n, C.c_void_p) for n in ("obs_erase", "obs_negative_depth", "point_delete", "point_error", "summary")]
        out = dict(obs_erase=np.zeros(self.O, np.uint8), obs_negative_depth=np.zeros(self.O, np.uint8),
                   point_delete=np.zeros(self.P, np.uint8), point_error=np.zeros(self.P), summary=np.zeros(4))
        fo = FO(*[_vp(out[k]) for k in ("obs_erase", "obs_negative_depth", "point_delete", "point_error", "summary")])
        L = lib()
        L.pcd_ba_filter_tracks.argtypes = [C.c_void_p, C.c_double, C.c_void_p]
        _check(L.pcd_ba_filter_tracks(self._h, float(max_reproj_error), C.byref(fo)))
        sm = out.pop("summary")
        out.update(num_filtered=int(sm[0]), mean_reproj_error=float(sm[1]), num_points_with_error=int(sm[2]),
                   num_negative_depth=int(sm[3]))
        return out

    def device_parameters(self):
        a, b = C.c_void_p(), C.c_void_p()
        _check(lib().pcd_ba_device_parameters(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value
