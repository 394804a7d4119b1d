"""CPU test pinning the oracle's scan-to-scan data association (odometry_scan_matcher.cc:81-258) with the loops of
the reference written out once more in plain Python, independently of oracle/msfl_oracle.c: f64 transform of the
query cast to f32, exact 1-NN by f32 squared distance (what the kd-tree returns), the index-ordered forward / backward
sweeps with their `continue` / `break` rules and strict '<' running minima, then the line / plane construction of
lidar_factor.h:70-78."""
import numpy as np

from msf_loam_amd import synth
from tests import common

TH, NEAR = 25.0, 2.5


def _transform(pose, p):
    """TransformToStart with s = 1: Identity.slerp(1, q) * p + t in f64, cast to f32."""
    t, q = pose[:3], pose[3:]
    R = synth.quat_to_matrix(q)
    return (R @ p.astype(np.float64) + t).astype(np.float32)


def _d2(a, b):
    """(ax-bx)*(ax-bx) + (ay-by)*(ay-by) + (az-bz)*(az-bz), every operation in f32."""
    d = (a - b).astype(np.float32)
    s = (d * d).astype(np.float32)
    return np.float32(np.float32(s[..., 0] + s[..., 1]) + s[..., 2])


def _py_associate(ls, ls_ring, lf, lf_ring, sharp, flat, pose):
    out = []
    ls3, lf3 = ls[:, :3].astype(np.float32), lf[:, :3].astype(np.float32)
    for p in sharp[:, :3]:
        sel = _transform(pose, p)
        rec = (0, np.zeros(3), np.zeros(3))
        if len(ls3):
            d = _d2(ls3, sel)
            c = int(np.argmin(d))                                   # first minimum: the lowest index among exact ties
            if np.float64(d[c]) < TH:
                rid, best, j2 = int(ls_ring[c]), TH, -1
                for j in range(c + 1, len(ls3)):
                    if ls_ring[j] <= rid: continue
                    if ls_ring[j] > rid + NEAR: break
                    if np.float64(d[j]) < best: best, j2 = np.float64(d[j]), j
                for j in range(c - 1, -1, -1):
                    if ls_ring[j] >= rid: continue
                    if ls_ring[j] < rid - NEAR: break
                    if np.float64(d[j]) < best: best, j2 = np.float64(d[j]), j
                if j2 >= 0:
                    a, b = ls3[c].astype(np.float64), ls3[j2].astype(np.float64)
                    n = a - b
                    rec = (1, a, n / np.sqrt(n @ n))
        out.append((p.astype(np.float64),) + rec)
    for p in flat[:, :3]:
        sel = _transform(pose, p)
        rec = (0, np.zeros(3), np.zeros(3))
        if len(lf3):
            d = _d2(lf3, sel)
            c = int(np.argmin(d))
            if np.float64(d[c]) < TH:
                rid, b2, b3, j2, j3 = int(lf_ring[c]), TH, TH, -1, -1
                for j in range(c + 1, len(lf3)):
                    if lf_ring[j] > rid + NEAR: break
                    if lf_ring[j] <= rid and np.float64(d[j]) < b2: b2, j2 = np.float64(d[j]), j
                    elif lf_ring[j] > rid and np.float64(d[j]) < b3: b3, j3 = np.float64(d[j]), j
                for j in range(c - 1, -1, -1):
                    if lf_ring[j] < rid - NEAR: break
                    if lf_ring[j] >= rid and np.float64(d[j]) < b2: b2, j2 = np.float64(d[j]), j
                    elif lf_ring[j] < rid and np.float64(d[j]) < b3: b3, j3 = np.float64(d[j]), j
                if j2 >= 0 and j3 >= 0:
                    a, b, cc = (lf3[k].astype(np.float64) for k in (c, j2, j3))
                    n = np.cross(a - b, a - cc)
                    rec = (2, (a + b + cc) / 3, n / np.sqrt(n @ n))
        out.append((p.astype(np.float64),) + rec)
    return out


def _pair(seed, n_az=400):
    w, _, _ = common.small_world(20000)
    rng = np.random.default_rng(seed)
    pose = synth.random_poses(1, synth.SEED + 300 + seed)[0]
    nxt = synth.perturb_pose(pose, rng, 0.3, 3.0)
    return w, pose, nxt, n_az


def test_association_equals_a_plain_python_restatement(oracle):
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    n_edge = n_plane = 0
    for seed in range(3):
        w, pose, nxt, n_az = _pair(seed)
        fa = oracle.extract_features(*synth.make_scan(w, pose, synth.SEED + 310 + seed, n_az=n_az))
        fb = oracle.extract_features(*synth.make_scan(w, nxt, synth.SEED + 320 + seed, n_az=n_az))
        ls, lsr = fa["full"][fa["less_sharp"]], fa["ring"][fa["less_sharp"]]
        lf, lfr = fa["full"][fa["less_flat"]], fa["ring"][fa["less_flat"]]
        sharp, flat = fb["full"][fb["sharp"]], fb["full"][fb["flat"]]
        if seed == 2:                                                  # a block out of ring order: the `break`s cut the sweeps short
            k = len(lf) // 3
            order = np.concatenate([np.arange(k, 2 * k), np.arange(0, k), np.arange(2 * k, len(lf))])
            lf, lfr = lf[order], lfr[order]
        guess = ident if seed != 1 else np.array([0.4, -0.3, 0.05, 0, 0, np.sin(0.02), np.cos(0.02)])
        got = oracle.associate_scan2scan(ls, lsr, lf, lfr, sharp, flat, guess)
        want = _py_associate(ls, lsr, lf, lfr, sharp, flat, guess)
        assert len(got) == len(want)
        for g, (p, kind, C, N) in zip(got, want):
            assert int(g["kind"]) == kind
            assert np.array_equal(g["p"], p)
            if kind:
                assert np.allclose(g["C"], C, rtol=0, atol=1e-12) and np.allclose(g["N"], N, rtol=0, atol=1e-12)
        n_edge += sum(1 for x in want if x[1] == 1); n_plane += sum(1 for x in want if x[1] == 2)
    assert n_edge > 100 and n_plane > 300
