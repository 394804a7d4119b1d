"""N3: GetDeltaQP / IMU deskew / TransformPointCloud — oracle known answers on the
CPU, GPU-vs-oracle parity through the C ABI on the MI355X."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation, Slerp

from msf_loam_amd import capi, synth
from tests import common


def _preintegration(n=45, span=0.1, seed=3):
    """A scan's worth of IMU pre-integration: constant body rate + acceleration, irregular sample times."""
    rng = np.random.default_rng(seed)
    t = np.sort(np.concatenate([[0.0, span], rng.uniform(0, span, n - 2)]))
    omega = np.array([0.4, -0.25, 1.3])
    acc = np.array([1.5, -0.7, 0.3])
    dq = Rotation.from_rotvec(omega[None, :] * t[:, None]).as_quat()          # [x y z w]
    dp = 0.5 * acc[None, :] * t[:, None] ** 2
    return t, dq, dp


def _cloud(n=5000, span=0.1, seed=4):
    rng = np.random.default_rng(seed)
    pts = np.zeros((n, 4), np.float32)
    pts[:, :3] = rng.uniform(-40, 40, (n, 3))
    pts[:, 3] = rng.uniform(0, span, n).astype(np.float32)
    top = np.float32(span)                                    # the largest f32 time still inside the f64 span
    pts[0, 3], pts[1, 3] = 0.0, (top if float(top) <= span else np.nextafter(top, np.float32(0)))
    return pts


# ---------------------------------------------------------------- oracle known answers (CPU)

def test_delta_qp_matches_scipy_slerp_and_lerp(oracle):
    t, dq, dp = _preintegration()
    ref = Slerp(t, Rotation.from_quat(dq))
    for dt in np.linspace(0.0, 0.1, 37)[:-1]:
        rc, q, p = oracle.delta_qp(t, dq, dp, dt)
        assert rc == 0
        qr = ref([dt]).as_quat()[0]
        if np.dot(qr, q) < 0:
            qr = -qr
        assert np.max(np.abs(q - qr)) < 1e-12
        assert abs(np.linalg.norm(q) - 1.0) < 1e-12              # unit inputs stay unit to rounding (no normalisation step)
        assert np.max(np.abs(p - np.array([np.interp(dt, t, dp[:, k]) for k in range(3)]))) < 1e-15


def test_delta_qp_nodes_range_and_last_sample(oracle):
    t, dq, dp = _preintegration()
    for i in (0, 7, 20, 43):
        rc, q, p = oracle.delta_qp(t, dq, dp, t[i])
        assert rc == 0 and np.max(np.abs(q - dq[i])) < 1e-15 and np.array_equal(p, dp[i])
    rc, q, p = oracle.delta_qp(t, dq, dp, t[-1])                 # reference reads past the end here; restated as s = 1
    assert rc == 0 and np.max(np.abs(q - dq[-1])) < 1e-15 and np.max(np.abs(p - dp[-1])) < 1e-18
    assert oracle.delta_qp(t, dq, dp, -1e-9)[0] == 1             # CHECK(dt >= front) (:26)
    assert oracle.delta_qp(t, dq, dp, t[-1] + 1e-9)[0] == 1      # CHECK(dt <= back)
    assert oracle.delta_qp(t, dq, dp, float("nan"))[0] == 1


def test_slerp_takes_the_short_arc_and_the_linear_branch(oracle):
    t = np.array([0.0, 1.0])
    a = Rotation.from_rotvec([0, 0, 0.3]).as_quat()
    b = -Rotation.from_rotvec([0, 0, 0.5]).as_quat()             # same rotation, opposite sign: d < 0 flips the weight
    rc, q, _ = oracle.delta_qp(t, np.stack([a, b]), np.zeros((2, 3)), 0.5)
    assert np.max(np.abs(q - Rotation.from_rotvec([0, 0, 0.4]).as_quat())) < 1e-15
    rc, q, _ = oracle.delta_qp(t, np.stack([a, a]), np.zeros((2, 3)), 0.25)     # |dot| >= 1 - eps: linear blend
    assert np.max(np.abs(q - a)) < 1e-16


def test_deskew_cloud_formula(oracle):
    t, dq, dp = _preintegration()
    pts = _cloud(400)
    rot = Rotation.from_rotvec([0.1, -0.2, 0.7]).as_quat()
    v, g = np.array([3.0, -1.0, 0.2]), np.array([0.05, -0.1, 9.8])
    bad, out = oracle.deskew_cloud(t, dq, dp, pts, rot, v, g)
    assert bad == 0 and np.array_equal(out[:, 3], pts[:, 3])
    ref = Slerp(t, Rotation.from_quat(dq))
    for i in range(0, 400, 37):
        dt = float(pts[i, 3])
        e = ref([dt]).apply(pts[i, :3].astype(np.float64))[0] + Rotation.from_quat(rot).inv().apply(v * dt - 0.5 * g * dt * dt) + \
            np.array([np.interp(dt, t, dp[:, k]) for k in range(3)])
        assert np.max(np.abs(out[i, :3] - e)) <= 4e-6
    ident = np.tile([0, 0, 0, 1.0], (len(t), 1))
    bad, same = oracle.deskew_cloud(t, ident, np.zeros_like(dp), pts, rot, np.zeros(3), np.zeros(3))
    assert bad == 0 and np.array_equal(same, pts)
    pts[5, 3] = 0.2
    assert oracle.deskew_cloud(t, dq, dp, pts, rot, v, g)[0] == 1


def test_transform_cloud_is_transform_point(oracle):
    pts = _cloud(300)
    pose = synth.random_poses(1, 5)[0]
    out = oracle.transform_cloud(pts, pose)
    R = Rotation.from_quat(pose[3:]).as_matrix()
    ref = (pts[:, :3].astype(np.float64) @ R.T + pose[:3]).astype(np.float32)
    assert np.max(np.abs(out[:, :3] - ref)) <= 8e-6 and np.array_equal(out[:, 3], pts[:, 3])


# ---------------------------------------------------------------- GPU parity

@pytest.mark.gpu
def test_gpu_delta_qp_and_cloud_passes(gpu, oracle):
    t, dq, dp = _preintegration()
    pts = _cloud(30000)
    s, gq, gp = gpu.delta_qp(t, dq, dp, pts)
    assert s == 0
    for i in range(0, len(pts), 997):
        rc, q, p = oracle.delta_qp(t, dq, dp, float(pts[i, 3]))
        assert rc == 0 and np.max(np.abs(gq[i] - q)) < 1e-13 and np.max(np.abs(gp[i] - p)) < 1e-16
    rot = Rotation.from_rotvec([0.1, -0.2, 0.7]).as_quat()
    v, g = np.array([3.0, -1.0, 0.2]), np.array([0.05, -0.1, 9.8])
    s, out = gpu.deskew_cloud(t, dq, dp, pts, rot, v, g)
    bad, ref = oracle.deskew_cloud(t, dq, dp, pts, rot, v, g)
    assert s == 0 and bad == 0
    # acos/sin differ from glibc by an ulp of f64: after the cast to f32 at most a last-bit flip, and rarely
    ulp = np.abs(out[:, :3].view(np.int32).astype(np.int64) - ref[:, :3].view(np.int32).astype(np.int64))
    assert ulp.max() <= 1 and (ulp > 0).mean() < 1e-3
    s, out = gpu.undistort_cloud(t, dq, dp, pts)
    bad, ref = oracle.undistort_cloud(t, dq, dp, pts)
    assert s == 0 and bad == 0
    ulp = np.abs(out[:, :3].view(np.int32).astype(np.int64) - ref[:, :3].view(np.int32).astype(np.int64))
    assert ulp.max() <= 2 and (ulp > 0).mean() < 1e-2
    pose = synth.random_poses(1, 6)[0]
    assert np.array_equal(gpu.transform_cloud(pts, pose), oracle.transform_cloud(pts, pose))      # pure f64 mul/add: bit-exact


@pytest.mark.gpu
def test_gpu_time_outside_the_preintegration_is_an_error(gpu):
    t, dq, dp = _preintegration()
    pts = _cloud(100)
    pts[40, 3] = 0.1001
    assert gpu.delta_qp(t, dq, dp, pts)[0] == capi.BAD_ARG
    assert gpu.deskew_cloud(t, dq, dp, pts, [0, 0, 0, 1.0], np.zeros(3), np.zeros(3))[0] == capi.BAD_ARG
    pts[40, 3] = -0.001
    assert gpu.undistort_cloud(t, dq, dp, pts)[0] == capi.BAD_ARG
    assert gpu.delta_qp(t[:1], dq[:1], dp[:1], pts[:10])[0] == capi.BAD_ARG          # < 2 samples


@pytest.mark.gpu
def test_gpu_delta_qp_feeds_the_deskew_matcher(gpu, oracle):
    """GetDeltaQP on the device -> msfl_match_scan2map_deskew == the oracle fed by its own GetDeltaQP."""
    w, mc, ms = common.small_world()
    truth = synth.random_poses(4, synth.SEED + 31)[2]
    pts, ring = synth.make_scan(w, truth, 4242)
    f = oracle.extract_features(pts, ring)
    corner = oracle.voxel_grid(f["full"][f["less_sharp"]], 0.2)
    surf = oracle.voxel_grid(f["full"][f["less_flat"]], 0.4)
    t, dq, dp = _preintegration()
    dq[:] = Rotation.from_rotvec(np.array([0.02, -0.01, 0.05])[None, :] * t[:, None] / 0.1).as_quat()     # a gentle motion
    dp[:] = np.array([0.03, 0.01, 0.0])[None, :] * (t[:, None] / 0.1) ** 2
    for c in (corner, surf):
        c[:, 3] = np.clip(c[:, 3], 0.0, np.float32(0.0999))
    vel, grav = np.array([0.3, 0.1, 0.0]), np.array([0.0, 0.0, 9.8])
    guess = synth.perturb_pose(truth, np.random.default_rng(8), 0.1, 1.0)
    s1, cq, cp = gpu.delta_qp(t, dq, dp, corner)
    s2, sq, sp = gpu.delta_qp(t, dq, dp, surf)
    assert s1 == 0 and s2 == 0
    gpu.set_map(mc, ms)
    s, pose_g, _ = gpu.match_scan2map_deskew(corner, surf, cq, cp, sq, sp, vel, grav, guess)
    oq = np.array([oracle.delta_qp(t, dq, dp, float(x))[1] for x in corner[:, 3]])
    op = np.array([oracle.delta_qp(t, dq, dp, float(x))[2] for x in corner[:, 3]])
    oq2 = np.array([oracle.delta_qp(t, dq, dp, float(x))[1] for x in surf[:, 3]])
    op2 = np.array([oracle.delta_qp(t, dq, dp, float(x))[2] for x in surf[:, 3]])
    rc, pose_o, _ = oracle.match_scan2map_deskew(mc, ms, corner, surf, oq, op, oq2, op2, vel, grav, guess)
    assert s == rc == 0
    dt_, dr_ = synth.pose_error(pose_g, pose_o)
    assert dt_ < 1e-7 and dr_ < 1e-7


@pytest.mark.gpu
def test_gpu_in_place_passes_are_all_or_nothing_on_device_buffers(gpu):
    """with MSFL_MEM_DEVICE a failing msfl_deskew_cloud / msfl_undistort_cloud must leave the caller's cloud
    untouched (the reference CHECK-aborts before publishing anything), not half-modified."""
    import ctypes as C
    import torch
    t, dq, dp = _preintegration()
    pts = _cloud(5000)
    pts[4321, 3] = 0.1001                                    # one time stamp outside the span, late in the cloud
    pre, keep = gpu._preintegration(t, dq, dp)
    d = torch.from_numpy(pts.copy()).cuda()
    r, v, g = (np.ascontiguousarray(a, np.float64) for a in ([0, 0, 0, 1.0], [0.3, 0.1, 0.0], [0, 0, 9.81]))
    s = gpu.lib.msfl_deskew_cloud(gpu.h, C.byref(pre), capi._vp(d), C.c_int(len(pts)), capi._vp(r), capi._vp(v), capi._vp(g), C.c_int(capi.MEM_DEVICE))
    assert s == capi.BAD_ARG and np.array_equal(d.cpu().numpy(), pts)
    s = gpu.lib.msfl_undistort_cloud(gpu.h, C.byref(pre), capi._vp(d), C.c_int(len(pts)), C.c_int(capi.MEM_DEVICE))
    assert s == capi.BAD_ARG and np.array_equal(d.cpu().numpy(), pts)
    pts[4321, 3] = 0.05                                       # valid now: the same buffer is transformed
    d = torch.from_numpy(pts.copy()).cuda()
    s = gpu.lib.msfl_deskew_cloud(gpu.h, C.byref(pre), capi._vp(d), C.c_int(len(pts)), capi._vp(r), capi._vp(v), capi._vp(g), C.c_int(capi.MEM_DEVICE))
    assert s == 0
    s2, host = gpu.deskew_cloud(t, dq, dp, pts, r, v, g)
    assert s2 == 0 and np.array_equal(d.cpu().numpy(), host)
