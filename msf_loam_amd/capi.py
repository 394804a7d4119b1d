"""ctypes binding of the C ABI (include/msfl_c_api.h) exported by msf_loam_amd/libmsfl_hip.so.

This is plumbing for the Python harness (tests, bench.py).  It never falls back to a CPU path:
if the shared library is missing or a call fails, it raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MSFL_LIB: load another build of the same library (A/B runs of kernel variants)
LIB_PATH = os.environ.get("MSFL_LIB") or os.path.join(_HERE, "libmsfl_hip.so")

OK, TOO_FEW_CORRESPONDENCES, MAP_TOO_SMALL, BAD_ARG, HIP_ERROR, BAD_RING, NO_MAP, CAPACITY = range(8)
MEM_HOST, MEM_DEVICE = 0, 1

EXPORTED = [
    "msfl_default_params", "msfl_api_version", "msfl_create", "msfl_destroy", "msfl_set_stream", "msfl_reset_stream",
    "msfl_synchronize", "msfl_status_string", "msfl_last_error", "msfl_set_timing", "msfl_get_timing",
    "msfl_set_map", "msfl_match_scan2map", "msfl_match_scan2map_batch", "msfl_match_pairs_batch", "msfl_match_scan2map_deskew",
    "msfl_match_scan2map_deskew_batch",
    "msfl_associate_scan2map", "msfl_solve_records",
    "msfl_match_scan2scan", "msfl_match_scan2scan_batch", "msfl_extract_features",
    "msfl_extract_features_batch", "msfl_voxel_downsample", "msfl_voxel_downsample_batch", "msfl_voxel_downsample_batch_pair",
    "msfl_transform_cloud",
    "msfl_delta_qp", "msfl_deskew_cloud", "msfl_undistort_cloud",
    "msfl_grid_create", "msfl_grid_destroy", "msfl_grid_insert_scan", "msfl_grid_get_surrounded", "msfl_grid_size", "msfl_grid_dump",
    "msfl_slam_default_config", "msfl_slam_create", "msfl_slam_destroy", "msfl_slam_add_scan", "msfl_slam_add_scan_imu", "msfl_slam_get_result", "msfl_slam_grids",
    "msfl_slam_last_error", "msfl_slam_get_clouds",
]


class Preintegration(C.Structure):
    _fields_ = [("sum_dt", C.POINTER(C.c_double)), ("delta_q", C.POINTER(C.c_double)), ("delta_p", C.POINTER(C.c_double)),
                ("n", C.c_int)]


class Params(C.Structure):
    _fields_ = [
        ("scan_period", C.c_double), ("min_range", C.c_double),
        ("curvature_threshold", C.c_float), ("neighbor_gap_sq", C.c_float),
        ("sectors_per_ring", C.c_int), ("max_sharp_per_sector", C.c_int),
        ("max_less_sharp_per_sector", C.c_int), ("max_flat_per_sector", C.c_int),
        ("odom_distance_sq_threshold", C.c_double), ("odom_nearby_scan", C.c_double),
        ("odom_min_correspondences", C.c_int),
        ("map_knn", C.c_int), ("map_knn_max_sq_dist", C.c_float),
        ("line_eigen_ratio", C.c_double), ("plane_tolerance", C.c_double),
        ("outer_iterations", C.c_int), ("max_lm_iterations", C.c_int), ("huber_delta", C.c_double),
        ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
        ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double), ("max_consecutive_invalid_steps", C.c_int),
    ]


class MatchInfo(C.Structure):
    _fields_ = [("status", C.c_int), ("n_edge", C.c_int * 2), ("n_plane", C.c_int * 2),
                ("lm_iterations", C.c_int * 2), ("lm_successful", C.c_int * 2),
                ("initial_cost", C.c_double * 2), ("final_cost", C.c_double * 2)]


class Timing(C.Structure):
    _fields_ = [("launches_assoc", C.c_int), ("ms_assoc", C.c_double),
                ("launches_solve", C.c_int), ("ms_solve", C.c_double),
                ("launches_index", C.c_int), ("ms_index", C.c_double),
                ("launches_extract", C.c_int), ("ms_extract", C.c_double),
                ("launches_odom", C.c_int), ("ms_odom", C.c_double),
                ("launches_fit", C.c_int), ("ms_fit", C.c_double), ("knn_candidates", C.c_ulonglong), ("knn_candidates_seeded", C.c_ulonglong),
                ("launches_assoc_seeded", C.c_int), ("ms_assoc_seeded", C.c_double)]


class Deskew(C.Structure):
    _fields_ = [("corner_dq", C.c_void_p), ("corner_dp", C.c_void_p), ("surf_dq", C.c_void_p),
                ("surf_dp", C.c_void_p), ("velocity", C.c_double * 3), ("gravity", C.c_double * 3)]


class DeskewBatch(C.Structure):
    _fields_ = [("corner_dq", C.c_void_p), ("corner_dp", C.c_void_p), ("surf_dq", C.c_void_p),
                ("surf_dp", C.c_void_p), ("velocity", C.c_void_p), ("gravity", C.c_double * 3)]


class RingCloud(C.Structure):
    _fields_ = [("pts", C.c_void_p), ("ring", C.c_void_p), ("n", C.c_int)]


class RingCloudBatch(C.Structure):
    _fields_ = [("pts", C.c_void_p), ("ring", C.c_void_p), ("off", C.c_void_p)]


class Features(C.Structure):
    _fields_ = [("full_pts", C.c_void_p), ("full_ring", C.c_void_p), ("curvature", C.c_void_p),
                ("label", C.c_void_p), ("sharp_idx", C.c_void_p), ("less_sharp_idx", C.c_void_p),
                ("flat_idx", C.c_void_p), ("less_flat_idx", C.c_void_p),
                ("n_full", C.c_int), ("n_sharp", C.c_int), ("n_less_sharp", C.c_int),
                ("n_flat", C.c_int), ("n_less_flat", C.c_int)]


class FeaturesBatch(C.Structure):
    _fields_ = [("full_pts", C.c_void_p), ("full_ring", C.c_void_p), ("curvature", C.c_void_p),
                ("label", C.c_void_p), ("sharp_idx", C.c_void_p), ("less_sharp_idx", C.c_void_p),
                ("flat_idx", C.c_void_p), ("less_flat_idx", C.c_void_p),
                ("n_full", C.c_void_p), ("n_sharp", C.c_void_p), ("n_less_sharp", C.c_void_p),
                ("n_flat", C.c_void_p), ("n_less_flat", C.c_void_p)]


class SlamConfig(C.Structure):
    _fields_ = [("map_resolution", C.c_float), ("leaf_corner", C.c_float), ("leaf_surf", C.c_float),
                ("min_map_corner", C.c_int), ("min_map_surf", C.c_int), ("max_scan_points", C.c_int), ("max_rings", C.c_int),
                ("pose_odom2map", C.c_double * 7), ("reference_quirks", C.c_int), ("keep_clouds", C.c_int)]


class SlamClouds(C.Structure):
    """msfl_slam_clouds: cloud_full_res after the IMU passes, its map-frame copy, ring ids and the four index lists (keep_clouds)."""
    _fields_ = [("full_scan", C.c_void_p), ("full_map", C.c_void_p), ("ring", C.c_void_p),
                ("sharp_idx", C.c_void_p), ("less_sharp_idx", C.c_void_p), ("flat_idx", C.c_void_p), ("less_flat_idx", C.c_void_p),
                ("n_full", C.c_int), ("n_sharp", C.c_int), ("n_less_sharp", C.c_int), ("n_flat", C.c_int), ("n_less_flat", C.c_int)]


class SlamImu(C.Structure):
    """msfl_slam_imu: the per-scan IMU inputs of LaserMapping::Run (laser_mapping.cc:170-176,197-211)."""
    _fields_ = [("pre", C.POINTER(Preintegration)), ("is_initialized", C.c_int), ("velocity", C.c_double * 3),
                ("gravity", C.c_double * 3), ("presolved_pose", C.c_double * 7)]


class SlamResult(C.Structure):
    _fields_ = [("pose_odom", C.c_double * 7), ("pose_map", C.c_double * 7), ("pose_curr2last", C.c_double * 7),
                ("pose_odom2map", C.c_double * 7), ("odometry", MatchInfo), ("mapping", MatchInfo),
                ("scan_index", C.c_int), ("status_extract", C.c_int), ("status_mapping", C.c_int),
                ("n_full", C.c_int), ("n_sharp", C.c_int), ("n_less_sharp", C.c_int), ("n_flat", C.c_int), ("n_less_flat", C.c_int),
                ("n_corner_ds", C.c_int), ("n_surf_ds", C.c_int), ("n_map_corner", C.c_int), ("n_map_surf", C.c_int),
                ("grid_corner", C.c_int * 8), ("grid_surf", C.c_int * 8), ("status_imu", C.c_int), ("status_insert", C.c_int),
                ("status_clouds", C.c_int), ("reserved_", C.c_int)]


class MsflError(RuntimeError):
    def __init__(self, status, what, detail=""):
        self.status = status
        super().__init__(f"{what}: status {status} ({status_string(status)}) {detail}")


_lib = None


def load():
    """Load libmsfl_hip.so.  Raises (never falls back) if the HIP extension is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        lib.msfl_status_string.restype = C.c_char_p
        lib.msfl_last_error.restype = C.c_char_p
        lib.msfl_last_error.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


def status_string(s):
    return load().msfl_status_string(C.c_int(int(s))).decode()


def default_params():
    p = Params()
    load().msfl_default_params(C.byref(p))
    return p


def _vp(x):
    """numpy array / int device pointer / torch tensor / None -> c_void_p"""
    if x is None:
        return C.c_void_p(None)
    if isinstance(x, int):
        return C.c_void_p(x)
    if isinstance(x, np.ndarray):
        return C.c_void_p(x.ctypes.data)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    raise TypeError(type(x))


def _pts(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1, 4))


class Handle:
    """Owns one msfl_handle (one HIP stream + device scratch)."""

    def __init__(self, device=0, params=None):
        self.lib = load()
        self.h = C.c_void_p()
        s = self.lib.msfl_create(C.byref(params) if params is not None else None, C.c_int(device), C.byref(self.h))
        if s != OK:
            raise MsflError(s, "msfl_create", "(no GPU / HIP runtime? there is no CPU fallback)")

    def close(self):
        if self.h:
            self.lib.msfl_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, s, what, allow=()):
        if s != OK and s not in allow:
            raise MsflError(s, what, self.lib.msfl_last_error(self.h).decode())
        return s

    # ---- plumbing ----
    def set_stream(self, stream_ptr):
        self._check(self.lib.msfl_set_stream(self.h, C.c_void_p(stream_ptr)), "msfl_set_stream")

    def reset_stream(self):
        self._check(self.lib.msfl_reset_stream(self.h), "msfl_reset_stream")

    def synchronize(self):
        self._check(self.lib.msfl_synchronize(self.h), "msfl_synchronize")

    def set_timing(self, on=True):
        self._check(self.lib.msfl_set_timing(self.h, C.c_int(int(on))), "msfl_set_timing")

    def get_timing(self, reset=True):
        t = Timing()
        self._check(self.lib.msfl_get_timing(self.h, C.byref(t), C.c_int(int(reset))), "msfl_get_timing")
        return t

    # ---- stage C ----
    def set_map(self, corner, surf, n_corner=None, n_surf=None, mem=MEM_HOST):
        if mem == MEM_HOST:
            corner, surf = _pts(corner), _pts(surf)
            n_corner, n_surf = len(corner), len(surf)
            self._keep = (corner, surf)
        self._check(self.lib.msfl_set_map(self.h, _vp(corner), C.c_int(n_corner), _vp(surf), C.c_int(n_surf),
                                          C.c_int(mem)), "msfl_set_map")

    def match_scan2map(self, corner, surf, pose, want_info=True, allow=()):
        corner, surf = _pts(corner), _pts(surf)
        pose = np.array(pose, dtype=np.float64)
        info = MatchInfo()
        s = self.lib.msfl_match_scan2map(self.h, _vp(corner), C.c_int(len(corner)), _vp(surf), C.c_int(len(surf)),
                                         _vp(pose), C.byref(info) if want_info else None, C.c_int(MEM_HOST))
        self._check(s, "msfl_match_scan2map", allow)
        return s, pose, info

    def match_scan2map_batch(self, corner, corner_off, surf, surf_off, poses, want_info=False):
        """Host-memory batch. Returns (poses (B,7), status (B,), info list or None)."""
        corner, surf = _pts(corner), _pts(surf)
        co = np.ascontiguousarray(corner_off, dtype=np.int32)
        so = np.ascontiguousarray(surf_off, dtype=np.int32)
        poses = np.array(poses, dtype=np.float64).reshape(-1, 7).copy()
        B = len(poses)
        status = np.zeros(B, np.int32)
        info = (MatchInfo * B)() if want_info else None
        s = self.lib.msfl_match_scan2map_batch(self.h, C.c_int(B), _vp(corner), _vp(co), _vp(surf), _vp(so), _vp(poses),
                                               _vp(status), info, C.c_int(MEM_HOST))
        self._check(s, "msfl_match_scan2map_batch")
        return poses, status, info

    def match_scan2map_batch_device(self, B, corner_ptr, corner_off, surf_ptr, surf_off, poses_ptr, status_ptr=None):
        """Device-resident batch (asynchronous on the handle's stream). Offsets are host arrays."""
        co = np.ascontiguousarray(corner_off, dtype=np.int32)
        so = np.ascontiguousarray(surf_off, dtype=np.int32)
        s = self.lib.msfl_match_scan2map_batch(self.h, C.c_int(B), _vp(corner_ptr), _vp(co), _vp(surf_ptr), _vp(so),
                                               _vp(poses_ptr), _vp(status_ptr), None, C.c_int(MEM_DEVICE))
        self._check(s, "msfl_match_scan2map_batch(device)")

    def match_scan2map_deskew_batch(self, corner, corner_off, surf, surf_off, corner_dq, corner_dp, surf_dq, surf_dp,
                                    velocity, gravity, poses, mem=MEM_HOST, status=None):
        """Deskew branch for B scans.  Host memory: numpy arrays in, (poses, status) out.  Device memory: pass
        tensors / pointers (poses updated in place, `status` a device int32 buffer or None)."""
        co = np.ascontiguousarray(corner_off, dtype=np.int32)
        so = np.ascontiguousarray(surf_off, dtype=np.int32)
        B = len(co) - 1
        d = DeskewBatch()
        if mem == MEM_HOST:
            corner, surf = _pts(corner), _pts(surf)
            keep = [np.ascontiguousarray(a, dtype=np.float64) for a in (corner_dq, corner_dp, surf_dq, surf_dp, velocity)]
            poses = np.array(poses, dtype=np.float64).reshape(-1, 7).copy()
            status = np.zeros(B, np.int32)
        else:
            keep = [corner_dq, corner_dp, surf_dq, surf_dp, velocity]
        d.corner_dq, d.corner_dp, d.surf_dq, d.surf_dp, d.velocity = (_vp(a).value for a in keep)
        d.gravity = (C.c_double * 3)(*[float(g) for g in gravity])
        s = self.lib.msfl_match_scan2map_deskew_batch(self.h, C.c_int(B), _vp(corner), _vp(co), _vp(surf), _vp(so), C.byref(d),
                                                      _vp(poses), _vp(status), None, C.c_int(mem))
        self._check(s, "msfl_match_scan2map_deskew_batch")
        return poses, status

    def associate_scan2map(self, corner, surf, pose):
        """One data-association pass at `pose`: (n_corner+n_surf, 6) records {C, N}."""
        corner, surf = _pts(corner), _pts(surf)
        rec = np.zeros((max(len(corner) + len(surf), 1), 6))
        pose = np.ascontiguousarray(pose, dtype=np.float64)
        self._check(self.lib.msfl_associate_scan2map(self.h, _vp(corner), C.c_int(len(corner)), _vp(surf),
                                                     C.c_int(len(surf)), _vp(pose), _vp(rec)), "msfl_associate_scan2map")
        return rec[:len(corner) + len(surf)]

    def solve_records(self, corner, surf, records, pose):
        corner, surf = _pts(corner), _pts(surf)
        rec = np.ascontiguousarray(records, dtype=np.float64).reshape(-1, 6)
        pose = np.array(pose, dtype=np.float64)
        info = MatchInfo()
        self._check(self.lib.msfl_solve_records(self.h, _vp(corner), C.c_int(len(corner)), _vp(surf), C.c_int(len(surf)),
                                                _vp(rec), _vp(pose), C.byref(info)), "msfl_solve_records")
        return pose, info

    def match_scan2map_deskew(self, corner, surf, corner_dq, corner_dp, surf_dq, surf_dp, velocity, gravity, pose):
        corner, surf = _pts(corner), _pts(surf)
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (corner_dq, corner_dp, surf_dq, surf_dp)]
        d = Deskew()
        d.corner_dq, d.corner_dp, d.surf_dq, d.surf_dp = (a.ctypes.data for a in arrs)
        d.velocity = (C.c_double * 3)(*velocity)
        d.gravity = (C.c_double * 3)(*gravity)
        pose = np.array(pose, dtype=np.float64)
        info = MatchInfo()
        s = self.lib.msfl_match_scan2map_deskew(self.h, _vp(corner), C.c_int(len(corner)), _vp(surf), C.c_int(len(surf)),
                                                C.byref(d), _vp(pose), C.byref(info))
        self._check(s, "msfl_match_scan2map_deskew")
        return s, pose, info

    # ---- stage B ----
    def match_scan2scan(self, last_ls, last_ls_ring, last_lf, last_lf_ring, sharp, flat, pose, allow=(TOO_FEW_CORRESPONDENCES,)):
        clouds = []
        keep = []
        for pts, ring in ((last_ls, last_ls_ring), (last_lf, last_lf_ring), (sharp, None), (flat, None)):
            p = _pts(pts)
            r = np.ascontiguousarray(ring if ring is not None else np.zeros(len(p)), dtype=np.uint16)
            keep.append((p, r))
            rc = RingCloud()
            rc.pts, rc.ring, rc.n = p.ctypes.data, r.ctypes.data, len(p)
            clouds.append(rc)
        pose = np.array(pose, dtype=np.float64)
        info = MatchInfo()
        s = self.lib.msfl_match_scan2scan(self.h, C.byref(clouds[0]), C.byref(clouds[1]), C.byref(clouds[2]),
                                          C.byref(clouds[3]), _vp(pose), C.byref(info), C.c_int(MEM_HOST))
        self._check(s, "msfl_match_scan2scan", allow)
        return s, pose, info

    def match_scan2scan_batch(self, clouds, poses, want_info=False):
        """clouds: 4 tuples (pts (n,4), ring (n,), off (B+1,)) for last_less_sharp, last_less_flat,
        curr_sharp, curr_flat (ring may be None for the curr sets)."""
        structs, keep = [], []
        for pts, ring, off in clouds:
            p = _pts(pts)
            r = np.ascontiguousarray(ring if ring is not None else np.zeros(len(p)), dtype=np.uint16)
            o = np.ascontiguousarray(off, dtype=np.int32)
            keep.append((p, r, o))
            rb = RingCloudBatch()
            rb.pts, rb.ring, rb.off = p.ctypes.data, r.ctypes.data, o.ctypes.data
            structs.append(rb)
        poses = np.array(poses, dtype=np.float64).reshape(-1, 7).copy()
        B = len(poses)
        status = np.zeros(B, np.int32)
        info = (MatchInfo * B)() if want_info else None
        s = self.lib.msfl_match_scan2scan_batch(self.h, C.c_int(B), C.byref(structs[0]), C.byref(structs[1]),
                                                C.byref(structs[2]), C.byref(structs[3]), _vp(poses), _vp(status), info,
                                                C.c_int(MEM_HOST))
        self._check(s, "msfl_match_scan2scan_batch")
        return poses, status, info

    # ---- stage A ----
    def extract_features(self, pts, ring, extrinsic=None, allow=()):
        pts = _pts(pts)
        ring = np.ascontiguousarray(ring, dtype=np.uint16)
        n = max(len(pts), 1)
        out = dict(full=np.zeros((n, 4), np.float32), ring=np.zeros(n, np.uint16), curvature=np.zeros(n, np.float32),
                   label=np.zeros(n, np.uint8), sharp=np.zeros(n, np.int32), less_sharp=np.zeros(n, np.int32),
                   flat=np.zeros(n, np.int32), less_flat=np.zeros(n, np.int32))
        f = Features()
        f.full_pts, f.full_ring, f.curvature, f.label = (out[k].ctypes.data for k in ("full", "ring", "curvature", "label"))
        f.sharp_idx, f.less_sharp_idx, f.flat_idx, f.less_flat_idx = (out[k].ctypes.data for k in ("sharp", "less_sharp", "flat", "less_flat"))
        ext = np.ascontiguousarray(extrinsic, dtype=np.float64) if extrinsic is not None else None
        s = self.lib.msfl_extract_features(self.h, _vp(pts), _vp(ring), C.c_int(len(pts)), _vp(ext), C.byref(f), C.c_int(MEM_HOST))
        self._check(s, "msfl_extract_features", allow)
        nf = f.n_full
        return dict(rc=s, full=out["full"][:nf], ring=out["ring"][:nf], curvature=out["curvature"][:nf],
                    label=out["label"][:nf], sharp=out["sharp"][:f.n_sharp].copy(),
                    less_sharp=out["less_sharp"][:f.n_less_sharp].copy(), flat=out["flat"][:f.n_flat].copy(),
                    less_flat=out["less_flat"][:f.n_less_flat].copy())

    def extract_features_batch(self, pts, ring, off):
        """Host-memory batch.  Returns a list of per-scan dicts like extract_features()."""
        pts = _pts(pts)
        ring = np.ascontiguousarray(ring, dtype=np.uint16)
        off = np.ascontiguousarray(off, dtype=np.int32)
        B = len(off) - 1
        n = max(len(pts), 1)
        out = dict(full=np.zeros((n, 4), np.float32), ring=np.zeros(n, np.uint16), curvature=np.zeros(n, np.float32),
                   label=np.zeros(n, np.uint8), sharp=np.zeros(n, np.int32), less_sharp=np.zeros(n, np.int32),
                   flat=np.zeros(n, np.int32), less_flat=np.zeros(n, np.int32))
        cnt = {k: np.zeros(max(B, 1), np.int32) for k in ("n_full", "n_sharp", "n_less_sharp", "n_flat", "n_less_flat")}
        f = FeaturesBatch()
        f.full_pts, f.full_ring, f.curvature, f.label = (out[k].ctypes.data for k in ("full", "ring", "curvature", "label"))
        f.sharp_idx, f.less_sharp_idx, f.flat_idx, f.less_flat_idx = (out[k].ctypes.data for k in ("sharp", "less_sharp", "flat", "less_flat"))
        f.n_full, f.n_sharp, f.n_less_sharp, f.n_flat, f.n_less_flat = (cnt[k].ctypes.data for k in ("n_full", "n_sharp", "n_less_sharp", "n_flat", "n_less_flat"))
        status = np.zeros(max(B, 1), np.int32)
        s = self.lib.msfl_extract_features_batch(self.h, C.c_int(B), _vp(pts), _vp(ring), _vp(off), C.byref(f), _vp(status),
                                                 C.c_int(MEM_HOST))
        self._check(s, "msfl_extract_features_batch")
        res = []
        for b in range(B):
            o = int(off[b])
            nf = int(cnt["n_full"][b])
            res.append(dict(rc=int(status[b]), full=out["full"][o:o + nf], ring=out["ring"][o:o + nf],
                            curvature=out["curvature"][o:o + nf], label=out["label"][o:o + nf],
                            sharp=out["sharp"][o:o + cnt["n_sharp"][b]].copy(),
                            less_sharp=out["less_sharp"][o:o + cnt["n_less_sharp"][b]].copy(),
                            flat=out["flat"][o:o + cnt["n_flat"][b]].copy(),
                            less_flat=out["less_flat"][o:o + cnt["n_less_flat"][b]].copy()))
        return res

    def voxel_downsample(self, pts, leaf):
        pts = _pts(pts)
        out = np.zeros((max(len(pts), 1), 4), np.float32)
        n_out = C.c_int(0)
        self._check(self.lib.msfl_voxel_downsample(self.h, _vp(pts), C.c_int(len(pts)), C.c_float(leaf), _vp(out),
                                                   C.byref(n_out), C.c_int(MEM_HOST)), "msfl_voxel_downsample")
        return out[:n_out.value].copy()

    def voxel_downsample_batch(self, pts, off, leaf, idx=None, count=None):
        """Host-memory batch: clouds pts[off[b] (+ idx)] -> (filtered points back to back, out_off (B+1))."""
        pts = _pts(pts)
        off = np.ascontiguousarray(off, np.int32)
        B = len(off) - 1
        out = np.zeros((max(int(off[-1] - off[0]), 1), 4), np.float32)
        out_off = np.zeros(B + 1, np.int32)
        idx_a = None if idx is None else np.ascontiguousarray(idx, np.int32)
        cnt_a = None if count is None else np.ascontiguousarray(count, np.int32)
        self._check(self.lib.msfl_voxel_downsample_batch(self.h, C.c_int(B), _vp(pts), _vp(idx_a) if idx_a is not None else None, _vp(off),
                                                         _vp(cnt_a) if cnt_a is not None else None, C.c_float(leaf), _vp(out), _vp(out_off),
                                                         C.c_int(MEM_HOST)), "msfl_voxel_downsample_batch")
        return out[:out_off[-1]].copy(), out_off

    def match_pairs_batch(self, map_corner, map_corner_off, map_surf, map_surf_off, corner, corner_off, surf, surf_off, poses, want_info=False):
        """P (map, scan) pairs with distinct maps in one call (host memory) -> (poses (P,7), status (P,), info list or None)."""
        mc, ms, c, s_ = _pts(map_corner), _pts(map_surf), _pts(corner), _pts(surf)
        offs = [np.ascontiguousarray(o, np.int32) for o in (map_corner_off, map_surf_off, corner_off, surf_off)]
        P = len(offs[0]) - 1
        poses = np.ascontiguousarray(np.array(poses, np.float64).reshape(P, 7))
        status = np.zeros(P, np.int32)
        info = (MatchInfo * P)() if want_info else None
        self._check(self.lib.msfl_match_pairs_batch(self.h, C.c_int(P), _vp(mc), _vp(offs[0]), _vp(ms), _vp(offs[1]), _vp(c), _vp(offs[2]),
                                                    _vp(s_), _vp(offs[3]), _vp(poses), _vp(status), info, C.c_int(MEM_HOST)), "msfl_match_pairs_batch")
        return poses, status, info

    def match_pairs_batch_device(self, P, d_map_corner, map_corner_off, d_map_surf, map_surf_off, d_corner, corner_off, d_surf, surf_off,
                                 d_poses, d_status):
        """The same with device-resident clouds / poses / status (torch tensors or raw pointers) and host offset arrays; asynchronous."""
        offs = [np.ascontiguousarray(o, np.int32) for o in (map_corner_off, map_surf_off, corner_off, surf_off)]
        self._keep_offs = offs
        self._check(self.lib.msfl_match_pairs_batch(self.h, C.c_int(P), _vp(d_map_corner), _vp(offs[0]), _vp(d_map_surf), _vp(offs[1]), _vp(d_corner),
                                                    _vp(offs[2]), _vp(d_surf), _vp(offs[3]), _vp(d_poses), _vp(d_status), None, C.c_int(MEM_DEVICE)),
                    "msfl_match_pairs_batch(device)")

    def transform_cloud(self, pts, pose7):
        """TransformPointCloud (laser_mapping.cc:24-31)."""
        pts = _pts(pts)
        pose7 = np.ascontiguousarray(pose7, np.float64)
        out = np.zeros_like(pts)
        self._check(self.lib.msfl_transform_cloud(self.h, _vp(pts), C.c_int(len(pts)), _vp(pose7), _vp(out), C.c_int(MEM_HOST)),
                    "msfl_transform_cloud")
        return out

    @staticmethod
    def _preintegration(sum_dt, delta_q, delta_p):
        keep = (np.ascontiguousarray(sum_dt, np.float64), np.ascontiguousarray(delta_q, np.float64).reshape(-1, 4),
                np.ascontiguousarray(delta_p, np.float64).reshape(-1, 3))
        pre = Preintegration(keep[0].ctypes.data_as(C.POINTER(C.c_double)), keep[1].ctypes.data_as(C.POINTER(C.c_double)),
                             keep[2].ctypes.data_as(C.POINTER(C.c_double)), len(keep[0]))
        return pre, keep

    def delta_qp(self, sum_dt, delta_q, delta_p, pts):
        """GetDeltaQP (scan_undistortion.cc:22-42) per point -> (status, dq (n,4), dp (n,3))."""
        pts = _pts(pts)
        pre, keep = self._preintegration(sum_dt, delta_q, delta_p)
        dq, dp = np.zeros((len(pts), 4)), np.zeros((len(pts), 3))
        s = self.lib.msfl_delta_qp(self.h, C.byref(pre), _vp(pts), C.c_int(len(pts)), _vp(dq), _vp(dp), C.c_int(MEM_HOST))
        return s, dq, dp

    def deskew_cloud(self, sum_dt, delta_q, delta_p, pts, rot_odom_xyzw, velocity, gravity):
        """laser_mapping.cc:197-211 -> (status, deskewed points)."""
        pts = _pts(pts).copy()
        pre, keep = self._preintegration(sum_dt, delta_q, delta_p)
        r, v, g = (np.ascontiguousarray(a, np.float64) for a in (rot_odom_xyzw, velocity, gravity))
        s = self.lib.msfl_deskew_cloud(self.h, C.byref(pre), _vp(pts), C.c_int(len(pts)), _vp(r), _vp(v), _vp(g), C.c_int(MEM_HOST))
        return s, pts

    def undistort_cloud(self, sum_dt, delta_q, delta_p, pts):
        """UndistortScanInternal (scan_undistortion.cc:5-19) -> (status, points)."""
        pts = _pts(pts).copy()
        pre, keep = self._preintegration(sum_dt, delta_q, delta_p)
        s = self.lib.msfl_undistort_cloud(self.h, C.byref(pre), _vp(pts), C.c_int(len(pts)), C.c_int(MEM_HOST))
        return s, pts


class Grid:
    """Device-resident local map store (HybridGrid): msfl_grid_* over a Handle's stream."""

    def __init__(self, handle, resolution=3.0, leaf=0.2):
        self.handle = handle
        self.lib = handle.lib
        self.g = C.c_void_p()
        handle._check(self.lib.msfl_grid_create(handle.h, C.c_float(resolution), C.c_float(leaf), C.byref(self.g)), "msfl_grid_create")

    def close(self):
        if self.g:
            self.lib.msfl_grid_destroy(self.g)
            self.g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def insert_scan(self, pts, allow=()):
        pts = _pts(pts)
        return self.handle._check(self.lib.msfl_grid_insert_scan(self.g, _vp(pts), C.c_int(len(pts)), C.c_int(MEM_HOST)),
                                  "msfl_grid_insert_scan", allow)

    def size(self):
        a, b = C.c_int(0), C.c_int(0)
        self.handle._check(self.lib.msfl_grid_size(self.g, C.byref(a), C.byref(b)), "msfl_grid_size")
        return a.value, b.value

    def get_surrounded(self, scan, pose):
        scan = _pts(scan)
        cap = max(self.size()[0], 1)
        out = np.zeros((cap, 4), np.float32)
        n_out = C.c_int(0)
        pose = np.ascontiguousarray(pose, dtype=np.float64)
        self.handle._check(self.lib.msfl_grid_get_surrounded(self.g, _vp(scan), C.c_int(len(scan)), _vp(pose), _vp(out), C.c_int(cap),
                                                             C.byref(n_out), C.c_int(MEM_HOST)), "msfl_grid_get_surrounded")
        return out[:n_out.value].copy()

    def get_surrounded_device(self, scan_ptr, n, pose, out_ptr, capacity):
        """Device-resident variant: returns the number of points written at out_ptr."""
        n_out = C.c_int(0)
        pose = np.ascontiguousarray(pose, dtype=np.float64)
        self.handle._check(self.lib.msfl_grid_get_surrounded(self.g, _vp(scan_ptr), C.c_int(n), _vp(pose), _vp(out_ptr), C.c_int(capacity),
                                                             C.byref(n_out), C.c_int(MEM_DEVICE)), "msfl_grid_get_surrounded(device)")
        return n_out.value

    def dump(self):
        cap = max(self.size()[0], 1)
        out = np.zeros((cap, 4), np.float32)
        n_out = C.c_int(0)
        self.handle._check(self.lib.msfl_grid_dump(self.g, _vp(out), C.c_int(cap), C.byref(n_out), C.c_int(MEM_HOST)), "msfl_grid_dump")
        return out[:n_out.value].copy()


class _BorrowedGrid(Grid):
    """A map store owned by a Slam pipeline (never destroyed from here)."""

    def __init__(self, lib, g, err):
        self.lib, self.g, self._err = lib, g, err
        self.handle = self

    def _check(self, status, what, allow=()):
        if status != OK and status not in allow:
            raise MsflError(status, what, self._err())
        return status

    def close(self):
        self.g = C.c_void_p()


class Slam:
    """Device-resident per-scan SLAM step (msfl_slam_*): raw scan in, poses out, everything in between stays in HBM."""

    def __init__(self, device=0, max_scan_points=28800, max_rings=16, pose_odom2map=None, params=None, **cfg):
        self.lib = load()
        self.lib.msfl_slam_last_error.restype = C.c_char_p
        self.lib.msfl_slam_last_error.argtypes = [C.c_void_p]
        c = SlamConfig()
        self.lib.msfl_slam_default_config(C.byref(c))
        c.max_scan_points, c.max_rings = int(max_scan_points), int(max_rings)
        self.max_scan_points = int(max_scan_points)
        if pose_odom2map is not None:
            for k in range(7):
                c.pose_odom2map[k] = float(pose_odom2map[k])
        for k, v in cfg.items():
            setattr(c, k, v)
        self.s = C.c_void_p()
        st = self.lib.msfl_slam_create(C.byref(params) if params is not None else None, C.byref(c), C.c_int(device), C.byref(self.s))
        if st != OK:
            raise MsflError(st, "msfl_slam_create", "no GPU / HIP runtime available (there is no CPU fallback)" if st == HIP_ERROR else "")
        self.n_scans = 0

    def _err(self):
        return (self.lib.msfl_slam_last_error(self.s) or b"").decode()

    def close(self):
        if self.s:
            self.lib.msfl_slam_destroy(self.s)
            self.s = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def make_imu(sum_dt, delta_q, delta_p, is_initialized=False, velocity=(0, 0, 0), gravity=(0, 0, 0), presolved_pose=None):
        """Build an msfl_slam_imu (returns (struct, keep-alive tuple)); sum_dt None = no IMU data for the scan."""
        imu = SlamImu()
        keep = None
        if sum_dt is not None:
            pre, arrays = Handle._preintegration(sum_dt, delta_q, delta_p)
            keep = (pre, arrays)
            imu.pre = C.pointer(pre)
        imu.is_initialized = 1 if is_initialized else 0
        for k in range(3):
            imu.velocity[k], imu.gravity[k] = float(velocity[k]), float(gravity[k])
        pp = presolved_pose if presolved_pose is not None else (0, 0, 0, 0, 0, 0, 1)
        for k in range(7):
            imu.presolved_pose[k] = float(pp[k])
        return imu, keep

    def add_scan(self, pts, ring, wait=True, imu=None):
        """Feed one scan (host arrays).  wait=True: returns this scan's SlamResult; False: enqueue only.
        imu: a dict of make_imu's arguments (msfl_slam_add_scan_imu), or None (LiDAR-only)."""
        pts = _pts(pts)
        ring = np.ascontiguousarray(ring, dtype=np.uint16)
        r = SlamResult() if wait else None
        if imu is None:
            st = self.lib.msfl_slam_add_scan(self.s, _vp(pts), _vp(ring), C.c_int(len(pts)), C.c_int(MEM_HOST), C.byref(r) if wait else None)
        else:
            im, keep = self.make_imu(**imu)
            st = self.lib.msfl_slam_add_scan_imu(self.s, _vp(pts), _vp(ring), C.c_int(len(pts)), C.c_int(MEM_HOST), C.byref(im),
                                                 C.byref(r) if wait else None)
            del keep
        if st != OK:
            raise MsflError(st, "msfl_slam_add_scan", self._err())
        self.n_scans += 1
        return r

    def add_scan_device(self, pts_ptr, ring_ptr, n, wait=True):
        r = SlamResult() if wait else None
        st = self.lib.msfl_slam_add_scan(self.s, _vp(pts_ptr), _vp(ring_ptr), C.c_int(int(n)), C.c_int(MEM_DEVICE), C.byref(r) if wait else None)
        if st != OK:
            raise MsflError(st, "msfl_slam_add_scan(device)", self._err())
        self.n_scans += 1
        return r

    def result(self, scan_index):
        r = SlamResult()
        st = self.lib.msfl_slam_get_result(self.s, C.c_int(int(scan_index)), C.byref(r))
        if st != OK:
            raise MsflError(st, "msfl_slam_get_result", self._err())
        return r

    def clouds(self, scan_index):
        """keep_clouds=1: the scan's data products as host arrays (msfl_slam_get_clouds, MSFL_MEM_HOST): dict with full_scan (n,4),
        full_map (n,4), ring (n,), sharp / less_sharp / flat / less_flat index arrays.  Only for one of the last two scans fed."""
        n = self.max_scan_points
        full_scan, full_map = np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)
        ring = np.zeros(n, np.uint16)
        idx = [np.zeros(n, np.int32) for _ in range(4)]
        c = SlamClouds()
        c.full_scan, c.full_map, c.ring = full_scan.ctypes.data, full_map.ctypes.data, ring.ctypes.data
        c.sharp_idx, c.less_sharp_idx, c.flat_idx, c.less_flat_idx = (a.ctypes.data for a in idx)
        st = self.lib.msfl_slam_get_clouds(self.s, C.c_int(int(scan_index)), C.byref(c), C.c_int(MEM_HOST))
        if st != OK:
            raise MsflError(st, "msfl_slam_get_clouds", self._err())
        return dict(full_scan=full_scan[:c.n_full], full_map=full_map[:c.n_full], ring=ring[:c.n_full], sharp=idx[0][:c.n_sharp],
                    less_sharp=idx[1][:c.n_less_sharp], flat=idx[2][:c.n_flat], less_flat=idx[3][:c.n_less_flat])

    def grids(self):
        a, b = C.c_void_p(), C.c_void_p()
        st = self.lib.msfl_slam_grids(self.s, C.byref(a), C.byref(b))
        if st != OK:
            raise MsflError(st, "msfl_slam_grids", self._err())
        return _BorrowedGrid(self.lib, a, self._err), _BorrowedGrid(self.lib, b, self._err)
