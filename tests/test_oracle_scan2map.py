"""CPU test pinning the oracle's scan-to-map data association (mapping_scan_matcher.cc:109-246, the branch without
IMU de-skewing) end to end with a plain numpy restatement that shares no code with oracle/msfl_oracle.c: TransformPoint
(f32 -> f64 -> q p + t -> f32, rigid_transform.h:132-138), the five nearest map points by f32 squared distance, the
`dist[4] < 1.0` gate, the line test on the covariance eigenvalues (largest > 3 x middle) with the point 0.1 along the
direction, the plane normal from the 5 x 3 least squares A n = -1 with the 0.2 m point-to-plane check about the centroid."""
import numpy as np

from msf_loam_amd import synth
from tests import common


def _transform(pose, p):
    return (synth.quat_to_matrix(pose[3:]) @ p.astype(np.float64) + pose[:3]).astype(np.float32)


def _knn5(cloud3, q):
    d = (cloud3 - q).astype(np.float32)
    s = (d * d).astype(np.float32)
    d2 = (np.float32(s[:, 0] + s[:, 1]) + s[:, 2]).astype(np.float32)
    idx = np.lexsort((np.arange(len(d2)), d2))[:5]
    return idx, d2[idx]


def _np_associate(map_c, map_s, corner, surf, pose):
    out = []
    mc3, ms3 = map_c[:, :3].astype(np.float32), map_s[:, :3].astype(np.float32)
    for p in corner[:, :3]:
        rec = (0, None, None)
        if len(mc3) >= 5:
            idx, d2 = _knn5(mc3, _transform(pose, p))
            if d2[4] < 1.0:
                A = mc3[idx].astype(np.float64)
                c = A.mean(axis=0)
                w, U = np.linalg.eigh((A - c).T @ (A - c))
                rec = (1, c, U[:, 2], (w[2] - 3 * w[1]) / w[2]) if w[2] > 3 * w[1] else (0, None, None, (3 * w[1] - w[2]) / w[2])
        out.append(rec)
    for p in surf[:, :3]:
        rec = (0, None, None)
        if len(ms3) >= 5:
            idx, d2 = _knn5(ms3, _transform(pose, p))
            if d2[4] < 1.0:
                A = ms3[idx].astype(np.float64)
                n = np.linalg.lstsq(A, -np.ones(5), rcond=None)[0]
                n /= np.sqrt(n @ n)
                c = A.mean(axis=0)
                dist = np.abs((A - c) @ n)
                if np.all(dist <= 0.2):
                    rec = (2, c, n, 0.2 - dist.max())
                else:
                    rec = (0, None, None, dist.max() - 0.2)
        out.append(rec)
    return out


def test_association_equals_a_numpy_restatement(oracle):
    w, mc, ms = common.small_world(20000)
    rng = np.random.default_rng(3)
    n_edge = n_plane = n_rejected = 0
    for case in range(2):
        pose = synth.random_poses(1, synth.SEED + 400 + case)[0]
        f = oracle.extract_features(*synth.make_scan(w, pose, synth.SEED + 410 + case, n_az=450))
        corner = oracle.voxel_grid(f["full"][f["less_sharp"]], 0.2)
        surf = oracle.voxel_grid(f["full"][f["less_flat"]], 0.4)[::3]
        guess = synth.perturb_pose(pose, rng, 0.2, 2.0)
        got = oracle.associate_scan2map(mc, ms, corner, surf, guess, use_kdtree=True)
        want = _np_associate(mc, ms, corner, surf, guess)
        assert len(got) == len(want) == len(corner) + len(surf)
        for g, wnt in zip(got, want):
            kind = wnt[0]
            margin = wnt[3] if len(wnt) > 3 else 1.0
            if margin < 1e-9:                      # a decision on the edge of its threshold: either answer is right
                continue
            assert int(g["kind"]) == kind
            if kind == 1:
                c, d = wnt[1], wnt[2]
                assert abs(abs(g["N"] @ d) - 1) < 1e-9                       # the line direction, up to the eigenvector's sign
                assert np.allclose(g["C"], c + 0.1 * np.asarray(g["N"]), rtol=0, atol=1e-10)     # point_a = centre + 0.1 direction
                n_edge += 1
            elif kind == 2:
                assert np.allclose(g["N"], wnt[2], rtol=0, atol=1e-8) and np.allclose(g["C"], wnt[1], rtol=0, atol=1e-11)
                n_plane += 1
            else:
                n_rejected += 1
    assert n_edge > 50 and n_plane > 500 and n_rejected > 20


def test_deskew_matcher_reduces_to_the_plain_one(oracle):
    """The is_initialized branch (LidarEdge/PlaneFactorDeskewSE3, mapping_scan_matcher.cc:118-124, lidar_factor.cc:46-100)
    transforms a point as T * Rigid{R^T (V dt - g dt^2 / 2) + delta_p, delta_q} * p.  Two reductions that involve no
    code of the de-skew factors on the other side: (1) identity deltas with V = g = 0 ARE the plain matcher; (2) one rigid
    (delta_q, delta_p) shared by all points with V = g = 0 is the plain matcher on the points moved by that rigid motion
    (up to their f32 rounding)."""
    w, mc, ms = common.small_world(20000)
    rng = np.random.default_rng(11)
    pose = synth.random_poses(1, synth.SEED + 500)[0]
    f = oracle.extract_features(*synth.make_scan(w, pose, synth.SEED + 510, n_az=450))
    corner = oracle.voxel_grid(f["full"][f["less_sharp"]], 0.2)
    surf = oracle.voxel_grid(f["full"][f["less_flat"]], 0.4)
    guess = synth.perturb_pose(pose, rng, 0.15, 1.5)
    zero3 = np.zeros(3)
    ident = lambda n: np.tile([0, 0, 0, 1.0], (n, 1))
    rc0, p0, i0 = oracle.match_scan2map(mc, ms, corner, surf, guess)
    rc1, p1, i1 = oracle.match_scan2map_deskew(mc, ms, corner, surf, ident(len(corner)), np.zeros((len(corner), 3)), ident(len(surf)),
                                               np.zeros((len(surf), 3)), zero3, zero3, guess)
    assert rc0 == rc1 == 0 and list(i0.n_edge) == list(i1.n_edge) and list(i0.n_plane) == list(i1.n_plane)
    assert max(synth.pose_error(p0, p1)) < 1e-12
    # (2) a shared rigid motion of the points
    dq = synth.quat_from_rotvec(np.array([0.01, -0.02, 0.03])); dp = np.array([0.05, -0.02, 0.01])
    R = synth.quat_to_matrix(dq)
    def moved(c):
        m = c.copy(); m[:, :3] = (c[:, :3].astype(np.float64) @ R.T + dp).astype(np.float32); return m
    rc2, p2, i2 = oracle.match_scan2map_deskew(mc, ms, corner, surf, np.tile(dq, (len(corner), 1)), np.tile(dp, (len(corner), 1)),
                                               np.tile(dq, (len(surf), 1)), np.tile(dp, (len(surf), 1)), zero3, zero3, guess)
    rc3, p3, i3 = oracle.match_scan2map(mc, ms, moved(corner), moved(surf), guess)
    assert rc2 == rc3 == 0
    assert abs(i2.n_plane[1] - i3.n_plane[1]) <= 3 and abs(i2.n_edge[1] - i3.n_edge[1]) <= 3      # f32 rounding may flip a gate
    dt, dr = synth.pose_error(p2, p3)
    assert dt < 2e-5 and dr < 2e-5, (dt, dr)
    assert max(synth.pose_error(p2, p0)) > 1e-3                                                   # and the motion does matter
    # (3) one time stamp tau for every point, velocity V and gravity g: every transformed point is shifted by the same
    # o = V tau - g tau^2 / 2 in the map frame, so the solve from (guess - o) ends at (plain result - o)
    tau = 0.05
    V, g = np.array([0.8, -0.3, 0.1]), np.array([0.0, 0.0, 9.81])
    o = V * tau - 0.5 * g * tau * tau
    ct, st = corner.copy(), surf.copy()
    ct[:, 3] = tau; st[:, 3] = tau
    g_shift = guess.copy(); g_shift[:3] -= o
    rc4, p4, i4 = oracle.match_scan2map_deskew(mc, ms, ct, st, ident(len(ct)), np.zeros((len(ct), 3)), ident(len(st)), np.zeros((len(st), 3)),
                                               V, g, g_shift)
    rc5, p5, i5 = oracle.match_scan2map(mc, ms, ct, st, guess)
    assert rc4 == rc5 == 0 and list(i4.n_edge) == list(i5.n_edge) and list(i4.n_plane) == list(i5.n_plane)
    back = p4.copy(); back[:3] += o
    assert max(synth.pose_error(back, p5)) < 1e-7
