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
