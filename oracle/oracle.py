"""ctypes loader for the CPU ORACLE (oracle/libmsfl_oracle.so).

TEST INFRASTRUCTURE ONLY — see oracle/msfl_oracle.h.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.  PARITY UNPINNED by the reference's own
tests; pinned by the numpy/scipy known-answer tests under tests/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

POINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("t", "<f4")])
CORR = np.dtype([("p", "<f8", 3), ("C", "<f8", 3), ("N", "<f8", 3), ("kind", "<i4"), ("pad", "<i4")])


class SolverOptions(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int), ("huber_delta", C.c_double),
                ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
                ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double), ("max_consecutive_invalid_steps", C.c_int)]


MAX_TRACE = 16


class SolveSummary(C.Structure):
    _fields_ = [("iterations", C.c_int), ("successful_steps", C.c_int), ("termination", C.c_int),
                ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("trace_cost", C.c_double * MAX_TRACE), ("trace_radius", C.c_double * MAX_TRACE),
                ("trace_rel_decrease", C.c_double * MAX_TRACE), ("trace_step_norm", C.c_double * MAX_TRACE),
                ("trace_accepted", C.c_int * MAX_TRACE)]


class MatchInfo(C.Structure):
    _fields_ = [("n_edge", C.c_int * 2), ("n_plane", C.c_int * 2), ("lm_iterations", C.c_int * 2),
                ("lm_successful", C.c_int * 2), ("initial_cost", C.c_double * 2), ("final_cost", C.c_double * 2)]


def build(force=False):
    so = os.path.join(_HERE, "libmsfl_oracle.so")
    src = os.path.join(_HERE, "msfl_oracle.c")
    hdr = os.path.join(_HERE, "msfl_oracle.h")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libmsfl_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_evaluate.restype = C.c_double
        _LIB.orc_kdtree_build.restype = C.c_void_p
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def as_points(a):
    """(n,4) float32 or structured POINT -> contiguous (n,4) float32."""
    a = np.asarray(a)
    if a.dtype == POINT:
        a = a.view(np.float32).reshape(-1, 4)
    return np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 4)


# ---- math -------------------------------------------------------------------------------------

def quat_rotate(q, v):
    out = np.zeros(3)
    lib().orc_quat_rotate(_p(np.ascontiguousarray(q, dtype=np.float64)), _p(np.ascontiguousarray(v, dtype=np.float64)), _p(out))
    return out


def quat_to_matrix(q):
    out = np.zeros(9)
    lib().orc_quat_to_matrix(_p(np.ascontiguousarray(q, dtype=np.float64)), _p(out))
    return out.reshape(3, 3)


def transform_point(pose, p):
    out = np.zeros(3, np.float32)
    lib().orc_transform_point(_p(np.ascontiguousarray(pose, dtype=np.float64)), _p(np.ascontiguousarray(p, dtype=np.float32)), _p(out))
    return out


def pose_plus(x, delta):
    out = np.zeros(7)
    lib().orc_pose_plus(_p(np.ascontiguousarray(x, dtype=np.float64)), _p(np.ascontiguousarray(delta, dtype=np.float64)), _p(out))
    return out


def pose_compose(a, b):
    out = np.zeros(7)
    lib().orc_pose_compose(_p(np.ascontiguousarray(a, dtype=np.float64)), _p(np.ascontiguousarray(b, dtype=np.float64)), _p(out))
    return out


def sym_eigen3(A):
    ev, V = np.zeros(3), np.zeros(9)
    lib().orc_sym_eigen3(_p(np.ascontiguousarray(A, dtype=np.float64)), _p(ev), _p(V))
    return ev, V.reshape(3, 3)


def lstsq_5x3(A, b):
    x = np.zeros(3)
    rank = lib().orc_lstsq_5x3(_p(np.ascontiguousarray(A, dtype=np.float64)), _p(np.ascontiguousarray(b, dtype=np.float64)), _p(x))
    return x, rank


def edge_fit(nbr, ratio=3.0):
    Cc, N = np.zeros(3), np.zeros(3)
    ok = lib().orc_edge_fit(_p(np.ascontiguousarray(nbr, dtype=np.float32)), C.c_double(ratio), _p(Cc), _p(N))
    return bool(ok), Cc, N


def plane_fit(nbr, tol=0.2):
    Cc, N = np.zeros(3), np.zeros(3)
    ok = lib().orc_plane_fit(_p(np.ascontiguousarray(nbr, dtype=np.float32)), C.c_double(tol), _p(Cc), _p(N))
    return bool(ok), Cc, N


def edge_factor(pose, p, Cc, N):
    r, J = np.zeros(3), np.zeros(21)
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (pose, p, Cc, N)]
    lib().orc_edge_factor(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(r), _p(J))
    return r, J.reshape(3, 7)


def plane_factor(pose, p, Cc, N):
    r, J = np.zeros(1), np.zeros(7)
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (pose, p, Cc, N)]
    lib().orc_plane_factor(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(r), _p(J))
    return r, J.reshape(1, 7)


# ---- kNN --------------------------------------------------------------------------------------

def knn_brute(cloud, q, k=5):
    cloud = as_points(cloud)
    idx, d2 = np.zeros(k, np.int32), np.zeros(k, np.float32)
    lib().orc_knn_brute(_p(cloud), C.c_int(len(cloud)), _p(np.ascontiguousarray(q, dtype=np.float32)), C.c_int(k), _p(idx), _p(d2))
    return idx, d2


class KdTree:
    def __init__(self, cloud):
        self.cloud = as_points(cloud)
        self.h = C.c_void_p(lib().orc_kdtree_build(_p(self.cloud), C.c_int(len(self.cloud))))

    def knn(self, q, k=5):
        idx, d2 = np.zeros(k, np.int32), np.zeros(k, np.float32)
        lib().orc_kdtree_knn(self.h, _p(np.ascontiguousarray(q, dtype=np.float32)), C.c_int(k), _p(idx), _p(d2))
        return idx, d2

    def __del__(self):
        try:
            lib().orc_kdtree_free(self.h)
        except Exception:
            pass


# ---- solver -----------------------------------------------------------------------------------

def default_solver_options():
    o = SolverOptions()
    lib().orc_default_solver_options(C.byref(o))
    return o


def evaluate(corr, pose, huber=0.1, want_jac=True):
    corr = np.ascontiguousarray(corr, dtype=CORR)
    H, g = np.zeros(36), np.zeros(6)
    cost = lib().orc_evaluate(_p(corr), C.c_int(len(corr)), _p(np.ascontiguousarray(pose, dtype=np.float64)),
                              C.c_double(huber), _p(H) if want_jac else None, _p(g) if want_jac else None)
    return cost, H.reshape(6, 6), g


def ceres_solve(corr, pose, options=None):
    corr = np.ascontiguousarray(corr, dtype=CORR)
    pose = np.array(pose, dtype=np.float64)
    opt = options or default_solver_options()
    s = SolveSummary()
    lib().orc_ceres_solve(_p(corr), C.c_int(len(corr)), _p(pose), C.byref(opt), C.byref(s))
    return pose, s


# ---- stage C ----------------------------------------------------------------------------------

def associate_scan2map(map_corner, map_surf, corner, surf, pose, use_kdtree=True):
    mc, ms, c, s = (as_points(a) for a in (map_corner, map_surf, corner, surf))
    out = np.zeros(len(c) + len(s), dtype=CORR)
    lib().orc_associate_scan2map(_p(mc), C.c_int(len(mc)), _p(ms), C.c_int(len(ms)), _p(c), C.c_int(len(c)),
                                 _p(s), C.c_int(len(s)), _p(np.ascontiguousarray(pose, dtype=np.float64)),
                                 C.c_int(int(use_kdtree)), _p(out))
    return out


def match_scan2map(map_corner, map_surf, corner, surf, pose, use_kdtree=True):
    mc, ms, c, s = (as_points(a) for a in (map_corner, map_surf, corner, surf))
    pose = np.array(pose, dtype=np.float64)
    info = MatchInfo()
    rc = lib().orc_match_scan2map(_p(mc), C.c_int(len(mc)), _p(ms), C.c_int(len(ms)), _p(c), C.c_int(len(c)),
                                  _p(s), C.c_int(len(s)), _p(pose), C.c_int(int(use_kdtree)), C.byref(info))
    return rc, pose, info


def match_scan2map_batch(map_corner, map_surf, corner, corner_off, surf, surf_off, poses, threads=1,
                         rebuild_tree_per_scan=True):
    mc, ms, c, s = (as_points(a) for a in (map_corner, map_surf, corner, surf))
    co = np.ascontiguousarray(corner_off, dtype=np.int32)
    so = np.ascontiguousarray(surf_off, dtype=np.int32)
    poses = np.array(poses, dtype=np.float64).reshape(-1, 7).copy()
    B = len(poses)
    status = np.zeros(B, np.int32)
    lib().orc_match_scan2map_batch(_p(mc), C.c_int(len(mc)), _p(ms), C.c_int(len(ms)), C.c_int(B), _p(c), _p(co),
                                   _p(s), _p(so), _p(poses), _p(status), C.c_int(threads),
                                   C.c_int(int(rebuild_tree_per_scan)))
    return poses, status


def match_scan2map_batch_trees(tree_corner, tree_surf, corner, corner_off, surf, surf_off, poses, threads=1):
    """The batch against kd-trees (KdTree objects over the two map clouds) built by the caller beforehand."""
    mc, ms = tree_corner.cloud, tree_surf.cloud
    c, s = as_points(corner), as_points(surf)
    co = np.ascontiguousarray(corner_off, dtype=np.int32)
    so = np.ascontiguousarray(surf_off, dtype=np.int32)
    poses = np.array(poses, dtype=np.float64).reshape(-1, 7).copy()
    B = len(poses)
    status = np.zeros(B, np.int32)
    lib().orc_match_scan2map_batch_trees(_p(mc), C.c_int(len(mc)), tree_corner.h, _p(ms), C.c_int(len(ms)), tree_surf.h, C.c_int(B),
                                         _p(c), _p(co), _p(s), _p(so), _p(poses), _p(status), C.c_int(threads))
    return poses, status


def match_scan2map_batch_timed(map_corner, map_surf, corner, corner_off, surf, surf_off, poses):
    """Single-threaded batch with the reference's LOG_STEP_TIME stages: returns (poses, status,
    {"build tree", "Data association", "Solver time"} seconds summed over the scans)."""
    mc, ms, c, s = (as_points(a) for a in (map_corner, map_surf, corner, surf))
    co = np.ascontiguousarray(corner_off, dtype=np.int32)
    so = np.ascontiguousarray(surf_off, dtype=np.int32)
    poses = np.array(poses, dtype=np.float64).reshape(-1, 7).copy()
    B = len(poses)
    status = np.zeros(B, np.int32)
    st = np.zeros(3, np.float64)
    lib().orc_match_scan2map_batch_timed(_p(mc), C.c_int(len(mc)), _p(ms), C.c_int(len(ms)), C.c_int(B), _p(c), _p(co),
                                         _p(s), _p(so), _p(poses), _p(status), _p(st))
    return poses, status, {"build tree": float(st[0]), "Data association": float(st[1]), "Solver time": float(st[2])}


def match_scan2map_deskew(map_corner, map_surf, corner, surf, corner_dq, corner_dp, surf_dq, surf_dp,
                          velocity, gravity, pose):
    mc, ms, c, s = (as_points(a) for a in (map_corner, map_surf, corner, surf))
    f = [np.ascontiguousarray(a, dtype=np.float64) for a in (corner_dq, corner_dp, surf_dq, surf_dp, velocity, gravity)]
    pose = np.array(pose, dtype=np.float64)
    info = MatchInfo()
    rc = lib().orc_match_scan2map_deskew(_p(mc), C.c_int(len(mc)), _p(ms), C.c_int(len(ms)), _p(c), C.c_int(len(c)),
                                         _p(s), C.c_int(len(s)), _p(f[0]), _p(f[1]), _p(f[2]), _p(f[3]), _p(f[4]), _p(f[5]),
                                         _p(pose), C.byref(info))
    return rc, pose, info


# ---- stage B ----------------------------------------------------------------------------------

def _ring(a):
    return np.ascontiguousarray(a, dtype=np.uint16)


def associate_scan2scan(last_ls, last_ls_ring, last_lf, last_lf_ring, sharp, flat, pose):
    a, b, c, d = (as_points(x) for x in (last_ls, last_lf, sharp, flat))
    ra, rb = _ring(last_ls_ring), _ring(last_lf_ring)
    out = np.zeros(len(c) + len(d), dtype=CORR)
    lib().orc_associate_scan2scan(_p(a), _p(ra), C.c_int(len(a)), _p(b), _p(rb), C.c_int(len(b)), _p(c), C.c_int(len(c)),
                                  _p(d), C.c_int(len(d)), _p(np.ascontiguousarray(pose, dtype=np.float64)), _p(out))
    return out


def match_scan2scan(last_ls, last_ls_ring, last_lf, last_lf_ring, sharp, flat, pose):
    a, b, c, d = (as_points(x) for x in (last_ls, last_lf, sharp, flat))
    ra, rb = _ring(last_ls_ring), _ring(last_lf_ring)
    pose = np.array(pose, dtype=np.float64)
    info = MatchInfo()
    rc = lib().orc_match_scan2scan(_p(a), _p(ra), C.c_int(len(a)), _p(b), _p(rb), C.c_int(len(b)), _p(c), C.c_int(len(c)),
                                   _p(d), C.c_int(len(d)), _p(pose), C.byref(info))
    return rc, pose, info


# ---- stage A ----------------------------------------------------------------------------------

def extract_features(pts, ring, min_range=0.3, extrinsic=None):
    pts = as_points(pts)
    ring = _ring(ring)
    n = len(pts)
    full = np.zeros((max(n, 1), 4), np.float32)
    fring = np.zeros(max(n, 1), np.uint16)
    curv = np.zeros(max(n, 1), np.float32)
    label = np.zeros(max(n, 1), np.uint8)
    idx = [np.zeros(max(n, 1), np.int32) for _ in range(4)]
    counts = np.zeros(5, np.int32)
    ext = np.ascontiguousarray(extrinsic, dtype=np.float64) if extrinsic is not None else None
    rc = lib().orc_extract_features(_p(pts), _p(ring), C.c_int(n), C.c_double(min_range), _p(ext), _p(full), _p(fring),
                                    _p(curv), _p(label), _p(idx[0]), _p(idx[1]), _p(idx[2]), _p(idx[3]), _p(counts))
    nf = int(counts[0])
    return dict(rc=rc, full=full[:nf], ring=fring[:nf], curvature=curv[:nf], label=label[:nf],
                sharp=idx[0][:counts[1]].copy(), less_sharp=idx[1][:counts[2]].copy(),
                flat=idx[2][:counts[3]].copy(), less_flat=idx[3][:counts[4]].copy())


def voxel_grid(pts, leaf):
    pts = as_points(pts)
    out = np.zeros((max(len(pts), 1), 4), np.float32)
    m = lib().orc_voxel_grid(_p(pts), C.c_int(len(pts)), C.c_float(leaf), _p(out))
    return out[:m].copy()


# ---- N3: IMU deskew inputs ---------------------------------------------------------------------------

def _pre(sum_dt, delta_q, delta_p):
    return (np.ascontiguousarray(sum_dt, np.float64), np.ascontiguousarray(delta_q, np.float64).reshape(-1, 4),
            np.ascontiguousarray(delta_p, np.float64).reshape(-1, 3))


def delta_qp(sum_dt, delta_q, delta_p, dt):
    """GetDeltaQP for one time -> (rc, q[4], p[3])."""
    sd, dq, dp = _pre(sum_dt, delta_q, delta_p)
    q, p = np.zeros(4), np.zeros(3)
    rc = lib().orc_delta_qp(_p(sd), _p(dq), _p(dp), C.c_int(len(sd)), C.c_double(float(dt)), _p(q), _p(p))
    return rc, q, p


def delta_qp_cloud(sum_dt, delta_q, delta_p, pts):
    """GetDeltaQP at every point's relative time -> (refused, dq (n,4), dp (n,3))."""
    sd, dq, dp = _pre(sum_dt, delta_q, delta_p)
    pts = as_points(pts)
    oq, op = np.zeros((len(pts), 4)), np.zeros((len(pts), 3))
    bad = lib().orc_delta_qp_cloud(_p(sd), _p(dq), _p(dp), C.c_int(len(sd)), _p(pts), C.c_int(len(pts)), _p(oq), _p(op))
    return bad, oq, op


def deskew_cloud(sum_dt, delta_q, delta_p, pts, rot_odom, velocity, gravity):
    sd, dq, dp = _pre(sum_dt, delta_q, delta_p)
    pts = as_points(pts).copy()
    r, v, g = (np.ascontiguousarray(a, np.float64) for a in (rot_odom, velocity, gravity))
    bad = lib().orc_deskew_cloud(_p(sd), _p(dq), _p(dp), C.c_int(len(sd)), _p(pts), C.c_int(len(pts)), _p(r), _p(v), _p(g))
    return bad, pts


def undistort_cloud(sum_dt, delta_q, delta_p, pts):
    sd, dq, dp = _pre(sum_dt, delta_q, delta_p)
    pts = as_points(pts).copy()
    bad = lib().orc_undistort_cloud(_p(sd), _p(dq), _p(dp), C.c_int(len(sd)), _p(pts), C.c_int(len(pts)))
    return bad, pts


def transform_cloud(pts, pose):
    pts = as_points(pts)
    out = np.zeros_like(pts)
    pose = np.ascontiguousarray(pose, np.float64)
    lib().orc_transform_cloud(_p(pts), C.c_int(len(pts)), _p(pose), _p(out))
    return out


# ---- N1: HybridGrid local map store ---------------------------------------------------------------

class HybridGrid:
    """Oracle of src/slam/map/hybrid_grid.cc HybridGridImpl (InsertScan / GetSurroundedCloud)."""

    def __init__(self, resolution=3.0, leaf=0.2):
        lib().orc_grid_create.restype = C.c_void_p
        self.g = C.c_void_p(lib().orc_grid_create(C.c_float(resolution), C.c_float(leaf)))

    def insert_scan(self, pts):
        pts = as_points(pts)
        return lib().orc_grid_insert_scan(self.g, _p(pts), C.c_int(len(pts)))

    def get_surrounded(self, scan, pose):
        scan = as_points(scan)
        cap = max(self.size()[0], 1)
        out = np.zeros((cap, 4), np.float32)
        m = lib().orc_grid_get_surrounded(self.g, _p(scan), C.c_int(len(scan)), _p(np.ascontiguousarray(pose, dtype=np.float64)),
                                          _p(out), C.c_int(cap))
        return out[:m].copy()

    def size(self):
        nc = C.c_int(0)
        n = lib().orc_grid_size(self.g, C.byref(nc))
        return n, nc.value

    def dump(self):
        n = max(self.size()[0], 1)
        out = np.zeros((n, 4), np.float32)
        m = lib().orc_grid_dump(self.g, _p(out), C.c_int(n))
        return out[:m].copy()

    def __del__(self):
        try:
            lib().orc_grid_free(self.g)
        except Exception:
            pass
