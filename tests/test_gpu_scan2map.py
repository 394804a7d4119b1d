"""GPU parity of stage C (scan-to-map registration) against the CPU oracle, through the C ABI.

Tolerances: kNN / accept decisions are integer work -> identical accept sets; fitted {C, N} and
poses are f64 -> 1e-9 on records, 1e-4 m / 1e-4 rad on poses is the north-star bar; we assert the
much tighter 1e-7 that the implementation actually reaches.
"""
import os

import numpy as np
import pytest

from msf_loam_amd import synth
from tests import common

pytestmark = pytest.mark.gpu

POSE_TOL_M = 1e-4      # BASELINE.json north_star tolerance
POSE_TOL_RAD = 1e-4
TIGHT = 1e-7           # what f64-everywhere actually delivers


def _oracle_records(orc, mc, ms, corner, surf, pose):
    corr = orc.associate_scan2map(mc, ms, corner, surf, pose, use_kdtree=True)
    rec = np.zeros((len(corr), 6))
    ok = corr["kind"] != 0
    rec[ok, :3] = corr["C"][ok]
    rec[ok, 3:] = corr["N"][ok]
    return rec, corr


def test_association_matches_oracle(gpu, oracle):
    _, mc, ms = common.small_world()
    gpu.set_map(mc, ms)
    for pts, ring, truth, guess in common.scans(3):
        _, corner, surf = common.features_from_oracle(oracle, pts, ring)
        rec_g = gpu.associate_scan2map(corner, surf, guess)
        rec_o, corr = _oracle_records(oracle, mc, ms, corner, surf, guess)
        acc_g = np.any(rec_g[:, 3:] != 0, axis=1)
        acc_o = corr["kind"] != 0
        assert np.array_equal(acc_g, acc_o), "accepted-correspondence sets differ"
        assert acc_o.sum() > 1000
        # edge direction sign is arbitrary (eigenvector); compare up to sign, C via the line/plane it defines
        n_dot = np.abs(np.sum(rec_g[acc_o, 3:] * rec_o[acc_o, 3:], axis=1))
        assert np.all(np.abs(n_dot - 1) < 1e-9)
        nc = len(corner)
        pl = acc_o.copy(); pl[:nc] = False
        assert np.abs(rec_g[pl, :3] - rec_o[pl, :3]).max() < 1e-9
        ed = acc_o.copy(); ed[nc:] = False
        # C = center +- 0.1 dir: compare the center-line distance instead
        d = rec_g[ed, :3] - rec_o[ed, :3]
        perp = d - np.sum(d * rec_o[ed, 3:], axis=1, keepdims=True) * rec_o[ed, 3:]
        assert np.abs(perp).max() < 1e-9


def test_solver_matches_oracle_on_fixed_records(gpu, oracle):
    _, mc, ms = common.small_world()
    gpu.set_map(mc, ms)
    for pts, ring, truth, guess in common.scans(3):
        _, corner, surf = common.features_from_oracle(oracle, pts, ring)
        rec_o, corr = _oracle_records(oracle, mc, ms, corner, surf, guess)
        pose_o, summ = oracle.ceres_solve(corr, guess)
        pose_g, info = gpu.solve_records(corner, surf, rec_o, guess)
        dt, dr = synth.pose_error(pose_g, pose_o)
        assert dt < TIGHT and dr < TIGHT, (dt, dr)
        assert info.lm_iterations[0] == summ.iterations
        assert info.lm_successful[0] == summ.successful_steps
        assert abs(info.initial_cost[0] - summ.initial_cost) <= 1e-9 * summ.initial_cost
        assert abs(info.final_cost[0] - summ.final_cost) <= 1e-9 * summ.final_cost


def test_match_scan2map_pose_parity(gpu, oracle):
    _, mc, ms = common.small_world()
    gpu.set_map(mc, ms)
    for pts, ring, truth, guess in common.scans(4):
        _, corner, surf = common.features_from_oracle(oracle, pts, ring)
        rc, pose_o, info_o = oracle.match_scan2map(mc, ms, corner, surf, guess)
        s, pose_g, info_g = gpu.match_scan2map(corner, surf, guess)
        assert s == 0 and rc == 0
        dt, dr = synth.pose_error(pose_g, pose_o)
        assert dt < POSE_TOL_M and dr < POSE_TOL_RAD
        assert dt < TIGHT and dr < TIGHT, (dt, dr)
        assert list(info_g.n_edge) == list(info_o.n_edge)
        assert list(info_g.n_plane) == list(info_o.n_plane)
        assert list(info_g.lm_iterations) == list(info_o.lm_iterations)
        # and the registration actually registers: closer to truth than the guess
        assert synth.pose_error(pose_g, truth)[0] < 0.25 * synth.pose_error(guess, truth)[0] + 0.01


@pytest.mark.parametrize("seed", range(int(os.environ.get("MSFL_FUZZ_SEEDS", "6"))))
def test_randomised_maps_and_guesses_follow_the_oracle(gpu, oracle, seed):
    """Differential fuzzing of the hot path: the local map thinned at random and with a box cut out of it (sparse
    neighbourhoods, features with fewer than five map points inside the 1 m gate), feature lists thinned, the guess up
    to ~1 m / 3 degrees off (few accepted correspondences, the trust region shrinking, solves that stop early).  Status,
    accepted-correspondence counts and LM iteration counts of both outer iterations must equal the oracle's, the pose
    must agree to 1e-7, and the same scan in a batch must give the single call's bits."""
    rng = np.random.default_rng(7000 + seed)
    _, mc, ms = common.small_world()
    keep_c = rng.uniform(size=len(mc)) < rng.uniform(0.3, 1.0)
    keep_s = rng.uniform(size=len(ms)) < rng.uniform(0.15, 1.0)
    lo = rng.uniform(-15, 5, 3); hi = lo + rng.uniform(2, 12, 3)
    keep_s &= ~np.all((ms[:, :3] > lo) & (ms[:, :3] < hi), axis=1)
    mc2, ms2 = np.ascontiguousarray(mc[keep_c]), np.ascontiguousarray(ms[keep_s])
    pts, ring, truth, guess = common.scans(4)[seed % 4]
    _, corner, surf = common.features_from_oracle(oracle, pts, ring)
    corner = np.ascontiguousarray(corner[rng.uniform(size=len(corner)) < rng.uniform(0.3, 1.0)])
    surf = np.ascontiguousarray(surf[rng.uniform(size=len(surf)) < rng.uniform(0.2, 1.0)])
    g = np.array(truth, np.float64)
    g[:3] += rng.normal(0, rng.choice([0.05, 0.3, 0.6]), 3)
    axis = rng.normal(size=3); axis /= np.linalg.norm(axis)
    ang = rng.normal(0, rng.choice([0.005, 0.03]))
    dq = np.concatenate([np.sin(ang / 2) * axis, [np.cos(ang / 2)]])
    x, y, z, w = g[3:]; a, b, c, d = dq                                  # g.q * dq (xyzw)
    g[3:] = [w * a + x * d + y * c - z * b, w * b - x * c + y * d + z * a, w * c + x * b - y * a + z * d, w * d - x * a - y * b - z * c]
    gpu.set_map(mc2, ms2)
    rc, pose_o, info_o = oracle.match_scan2map(mc2, ms2, corner, surf, g)
    s, pose_g, info_g = gpu.match_scan2map(corner, surf, g)
    assert s == rc
    assert list(info_g.n_edge) == list(info_o.n_edge) and list(info_g.n_plane) == list(info_o.n_plane)
    assert list(info_g.lm_iterations) == list(info_o.lm_iterations)
    dt, dr = synth.pose_error(pose_g, pose_o)
    assert dt < TIGHT and dr < TIGHT, (dt, dr)
    co, so = [0, len(corner), 2 * len(corner)], [0, len(surf), 2 * len(surf)]
    poses, st, _ = gpu.match_scan2map_batch(np.concatenate([corner, corner]), co, np.concatenate([surf, surf]), so, [g, g])
    assert np.all(st == s) and np.array_equal(poses[0], pose_g) and np.array_equal(poses[1], pose_g)


def test_batch_equals_single_and_is_deterministic(gpu, oracle):
    _, mc, ms = common.small_world()
    gpu.set_map(mc, ms)
    cs, ss, co, so, guesses = [], [], [0], [0], []
    for pts, ring, truth, guess in common.scans(6):
        _, corner, surf = common.features_from_oracle(oracle, pts, ring)
        cs.append(corner); ss.append(surf)
        co.append(co[-1] + len(corner)); so.append(so[-1] + len(surf))
        guesses.append(guess)
    C, S = np.concatenate(cs), np.concatenate(ss)
    poses1, st1, _ = gpu.match_scan2map_batch(C, co, S, so, guesses)
    poses2, st2, _ = gpu.match_scan2map_batch(C, co, S, so, guesses)
    assert np.array_equal(poses1, poses2), "batched registration must be bitwise reproducible"
    assert np.all(st1 == 0)
    for b in range(len(guesses)):
        _, p, _ = gpu.match_scan2map(cs[b], ss[b], guesses[b])
        assert np.array_equal(p, poses1[b]), "batch result must equal the single-scan call bit for bit"


def test_host_batches_arrive_in_parts_and_pieces(gpu, oracle, monkeypatch):
    """A host-buffer batch crosses PCIe in parts (each registered completely while the next one is copied) that arrive in
    pieces (a part's first association pass follows them): whatever the split, every scan's result equals the unsplit
    host call bit for bit, also with ragged scans and a scan without features."""
    from msf_loam_amd import capi
    _, mc, ms = common.small_world()
    sc = common.scans(6)
    feats = [common.features_from_oracle(oracle, p, r)[1:] for p, r, _, _ in sc]
    corners, surfs, guesses = [], [], []
    for i in range(40):
        c, sf = feats[i % 6]
        if i == 17: c, sf = c[:0], sf[:0]                    # nothing to match
        if i % 5 == 3: c, sf = c[: len(c) // 2], sf[: len(sf) // 3]
        corners.append(c); surfs.append(sf); guesses.append(sc[i % 6][3])
    co = np.cumsum([0] + [len(c) for c in corners]).astype(np.int32)
    so = np.cumsum([0] + [len(x) for x in surfs]).astype(np.int32)
    C, S = np.concatenate(corners), np.concatenate(surfs)
    gpu.set_map(mc, ms)
    want, want_st, _ = gpu.match_scan2map_batch(C, co, S, so, guesses)          # 40 scans: below the default part size, one copy
    for part, pieces in ((8, 1), (8, 3), (20, 2), (13, 8)):
        monkeypatch.setenv("MSFL_H2D_CHUNK_SCANS", str(part))
        monkeypatch.setenv("MSFL_H2D_SUB_CHUNKS", str(pieces))
        h = capi.Handle(0)
        try:
            h.set_map(mc, ms)
            got, got_st, _ = h.match_scan2map_batch(C, co, S, so, guesses)
        finally:
            h.close()
        assert np.array_equal(got_st, want_st) and np.array_equal(got, want), (part, pieces)
    assert np.all(want_st == 0) and np.array_equal(want[17], np.asarray(guesses[17], np.float64))    # an empty problem leaves its pose alone


def test_edge_cases(gpu, oracle):
    from msf_loam_amd import capi
    _, mc, ms = common.small_world()
    gpu.set_map(mc, ms)
    pose = np.array([0, 0, 1.8, 0, 0, 0, 1.0])
    # empty scan: Ceres solves an empty problem, pose untouched (bit-exact)
    s, p, info = gpu.match_scan2map(np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32), pose)
    assert s == 0 and np.array_equal(p, pose)
    # features far away from the map: no neighbour within 1 m -> no residuals -> pose untouched
    far = np.zeros((50, 4), np.float32); far[:, 0] = 1e4
    s, p, info = gpu.match_scan2map(far, far, pose)
    assert s == 0 and np.array_equal(p, pose) and info.n_edge[0] == 0 and info.n_plane[0] == 0
    # un-normalised quaternion with no step taken round-trips bit-exactly (Rigid3(Vector7) does not normalise)
    pose2 = np.array([1, 2, 3, 0, 0, 0, 2.0])
    s, p, _ = gpu.match_scan2map(far, far, pose2)
    assert np.array_equal(p, pose2)
    # map too small
    h2 = capi.Handle(0)
    h2.set_map(mc[:3], ms)
    s, _, _ = h2.match_scan2map(far, far, pose, allow=(capi.MAP_TOO_SMALL,))
    assert s == capi.MAP_TOO_SMALL
    # no map at all
    h3 = capi.Handle(0)
    s, _, _ = h3.match_scan2map(far, far, pose, allow=(capi.NO_MAP,))
    assert s == capi.NO_MAP
    h2.close(); h3.close()


def test_deskew_variant_matches_oracle(gpu, oracle):
    """is_initialized branch (LidarEdge/PlaneFactorDeskewSE3, lidar_factor.cc:46-100) with synthetic
    per-point (delta_q, delta_p), velocity and gravity."""
    _, mc, ms = common.small_world()
    gpu.set_map(mc, ms)
    rng = np.random.default_rng(21)
    for pts, ring, truth, guess in common.scans(2):
        f, corner, surf = common.features_from_oracle(oracle, pts, ring)
        V = np.array([0.8, -0.3, 0.05]); G = np.array([0.0, 0.0, 9.81])
        def dqdp(cloud):
            t = cloud[:, 3].astype(np.float64)
            rv = np.outer(t, [0.02, -0.01, 0.3])                      # small rotation growing with time
            dq = np.stack([synth.quat_from_rotvec(r) for r in rv])
            dp = np.outer(t, [0.05, 0.02, -0.01]) + rng.normal(0, 1e-4, (len(t), 3))
            return dq, dp
        cdq, cdp = dqdp(corner); sdq, sdp = dqdp(surf)
        rc, pose_o, info_o = oracle.match_scan2map_deskew(mc, ms, corner, surf, cdq, cdp, sdq, sdp, V, G, guess)
        s, pose_g, info_g = gpu.match_scan2map_deskew(corner, surf, cdq, cdp, sdq, sdp, V, G, guess)
        assert s == 0 and rc == 0
        assert list(info_g.n_edge) == list(info_o.n_edge) and list(info_g.n_plane) == list(info_o.n_plane)
        dt, dr = synth.pose_error(pose_g, pose_o)
        assert dt < TIGHT and dr < TIGHT, (dt, dr)
        # zero deskew (identity dq, zero dp, V = G = 0) reproduces the plain matcher bit for bit
        zq = np.tile([0, 0, 0, 1.0], (len(corner), 1)); zs = np.tile([0, 0, 0, 1.0], (len(surf), 1))
        s, p0, _ = gpu.match_scan2map_deskew(corner, surf, zq, np.zeros((len(corner), 3)), zs, np.zeros((len(surf), 3)),
                                             np.zeros(3), np.zeros(3), guess)
        s, p1, _ = gpu.match_scan2map(corner, surf, guess)
        assert synth.pose_error(p0, p1)[0] < 1e-12


def test_deskew_batch_equals_single_calls_host_and_device(gpu, oracle):
    """msfl_match_scan2map_deskew_batch: per-scan velocities, per-feature (dq, dp) indexed like the features;
    every scan reproduces its single-scan call bit for bit, from host and from device memory."""
    import torch
    from msf_loam_amd import capi
    _, mc, ms = common.small_world()
    gpu.set_map(mc, ms)
    rng = np.random.default_rng(33)
    G = np.array([0.0, 0.0, 9.81])
    items = []
    for i, (pts, ring, truth, guess) in enumerate(common.scans(3)):
        _, corner, surf = common.features_from_oracle(oracle, pts, ring)
        V = np.array([0.8, -0.3, 0.05]) * (i + 1) / 2
        def dqdp(cloud, k=i):
            t = cloud[:, 3].astype(np.float64)
            dq = np.stack([synth.quat_from_rotvec(r) for r in np.outer(t, [0.02, -0.01, 0.1 * (k + 1)])])
            return dq, np.outer(t, [0.05, 0.02, -0.01]) + rng.normal(0, 1e-4, (len(t), 3))
        items.append((corner, surf, *dqdp(corner), *dqdp(surf), V, guess))
    co = np.cumsum([0] + [len(it[0]) for it in items]).astype(np.int32)
    so = np.cumsum([0] + [len(it[1]) for it in items]).astype(np.int32)
    cat = lambda k: np.concatenate([it[k] for it in items])
    guesses = np.stack([it[7] for it in items]); vel = np.stack([it[6] for it in items])
    poses, status = gpu.match_scan2map_deskew_batch(cat(0), co, cat(1), so, cat(2), cat(3), cat(4), cat(5), vel, G, guesses)
    assert np.all(status == 0)
    for b, it in enumerate(items):
        s, p, _ = gpu.match_scan2map_deskew(it[0], it[1], it[2], it[3], it[4], it[5], it[6], G, it[7])
        assert s == 0 and np.array_equal(p, poses[b]), b
    rc, pose_o, _ = oracle.match_scan2map_deskew(mc, ms, *items[1][:6], items[1][6], G, items[1][7])
    assert rc == 0 and max(synth.pose_error(poses[1], pose_o)) < TIGHT
    dev = torch.device("cuda", 0)
    t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (cat(0), cat(1), cat(2), cat(3), cat(4), cat(5), vel, guesses)]
    d_status = torch.zeros(3, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    gpu.match_scan2map_deskew_batch(t[0], co, t[1], so, t[2], t[3], t[4], t[5], t[6], G, t[7], mem=capi.MEM_DEVICE, status=d_status)
    gpu.synchronize()
    assert np.array_equal(t[7].cpu().numpy(), poses) and np.all(d_status.cpu().numpy() == 0)


def test_lattice_map_with_ties_and_duplicates(gpu, oracle):
    """Exact-kNN stress: a lattice map (many exactly equal f32 distances), duplicated points, queries on
    lattice nodes, on cell boundaries of the index and exactly on / just inside / just outside the
    d^2 < 1.0 gate.  The total order (distance, index) must reproduce the oracle's selection, so the
    fitted records agree."""
    rng = np.random.default_rng(77)
    g = np.arange(-6, 7, dtype=np.float32) * 0.5
    X, Y = np.meshgrid(g, g, indexing="ij")
    plane = np.stack([X.ravel(), Y.ravel(), np.full(X.size, -1.5, np.float32)], 1)      # z = -1.5 lattice, 0.5 m pitch
    wall = np.stack([np.full(X.size, 3.5, np.float32), X.ravel(), Y.ravel() + 1.5], 1)   # x = 3.5 lattice
    origin_plane = np.stack([X.ravel() + 20.0, Y.ravel(), np.zeros(X.size, np.float32)], 1)   # z = 0: n.p = -1 has no solution
    ms = np.concatenate([plane, wall, origin_plane, plane[::7]])                        # + exact duplicates
    ms = np.concatenate([ms, np.zeros((len(ms), 1), np.float32)], 1).astype(np.float32)
    ms = ms[rng.permutation(len(ms))]                                                   # index order != spatial order
    line = np.stack([np.zeros(60, np.float32), np.zeros(60, np.float32), np.arange(60, dtype=np.float32) * 0.125], 1)
    mc = np.concatenate([line, line[::5]])                                              # vertical pole with duplicates
    mc = np.concatenate([mc, np.zeros((len(mc), 1), np.float32)], 1).astype(np.float32)
    gpu.set_map(mc, ms)
    q = [[0, 0, -1.5], [0.25, 0.25, -1.5], [0.5, 0, -1.5], [1.0, 1.0, -1.5], [2.999, 0, -1.5], [3.25, 0.25, 1.5], [0, 0, -0.54],
         [0, 0, -0.5], [0, 0, -0.5001], [-3.0, -3.0, -1.5], [-3.5, 0, -1.5], [0.001, -0.001, -0.95], [3.5, 3.0, 4.5], [1.7, -2.2, -1.2],
         [20.0, 0.25, 0.1], [2.6, 0.1, 0.3], [3.5, -2.75, 1.25]]
    surf = np.array([p + [0.0] for p in q], np.float32)
    corner = np.array([[0.1, 0.0, 1.0, 0], [0.0, 0.0, 3.3, 0], [0.99, 0.0, 2.0, 0], [1.0, 0.0, 2.0, 0], [0.0, 0.05, 7.4, 0],
                       [0.0, 0.0, 8.4, 0]], np.float32)
    for pose in (np.array([0, 0, 0, 0, 0, 0, 1.0]), np.array([0.125, -0.25, 0.0, 0, 0, 0, 1.0])):
        rec = gpu.associate_scan2map(corner, surf, pose)
        corr = oracle.associate_scan2map(mc, ms, corner, surf, pose)
        ok_o = corr["kind"] != 0
        assert np.array_equal(np.any(rec[:, 3:] != 0, axis=1), ok_o)
        assert ok_o[len(corner):].sum() >= 8 and not ok_o[len(corner) + 14]      # the z = 0 lattice never yields a plane
        nc = len(corner)
        pl = ok_o.copy(); pl[:nc] = False
        if pl.any():
            assert np.max(np.abs(rec[pl, :3] - corr["C"][pl])) < 1e-9
            assert np.max(np.abs(rec[pl, 3:] - corr["N"][pl])) < 1e-9
        ed = ok_o.copy(); ed[nc:] = False
        assert ed.sum() >= 3
        n_dot = np.abs(np.sum(rec[ed, 3:] * corr["N"][ed], axis=1))            # eigenvector sign is free
        assert np.all(np.abs(n_dot - 1) < 1e-9)
        d = rec[ed, :3] - corr["C"][ed]
        perp = d - np.sum(d * corr["N"][ed], axis=1, keepdims=True) * corr["N"][ed]
        assert np.abs(perp).max() < 1e-9


def _synthetic_corr(rng, n_plane, n_edge, normals=None, noise=0.0, truth=None):
    """Hand-made correspondences around a known pose: kind 1 = edge {p, C, N}, kind 2 = plane."""
    from oracle.oracle import CORR
    truth = truth if truth is not None else np.array([0.3, -0.2, 0.1, 0, 0, np.sin(0.05), np.cos(0.05)])
    R = synth.quat_to_matrix(truth[3:])
    corr = np.zeros(n_edge + n_plane, dtype=CORR)
    for i in range(n_edge + n_plane):
        p = rng.uniform(-20, 20, 3).astype(np.float32).astype(np.float64)            # features are f32 points
        w = R @ p + truth[:3]
        if normals is None:
            N = rng.normal(size=3)
        else:
            N = np.array(normals[i % len(normals)], dtype=np.float64)
        N /= np.linalg.norm(N)
        corr[i]["p"], corr[i]["N"], corr[i]["kind"] = p, N, (1 if i < n_edge else 2)
        corr[i]["C"] = w + noise * rng.normal(size=3) + (N * rng.normal() if i < n_edge else 0)   # an edge point slides along N
    return corr, truth


@pytest.mark.parametrize("case", ["generic", "at_optimum", "edges_only", "two_normals", "outliers", "many_edges"])
def test_solver_corner_cases_follow_the_oracle(gpu, oracle, case):
    """The trust-region logic outside the comfortable regime: already converged, only edges, a
    rank-deficient geometry (two plane normals: the LM damping carries the solve), gross outliers
    (Huber).  Same iteration / acceptance counts and the same pose as the oracle."""
    rng = np.random.default_rng(5)
    kw = dict(generic=dict(n_plane=400, n_edge=40, noise=0.01),
              at_optimum=dict(n_plane=300, n_edge=30, noise=0.0),
              edges_only=dict(n_plane=0, n_edge=120, noise=0.01),
              two_normals=dict(n_plane=300, n_edge=0, noise=0.005, normals=[[0, 0, 1], [1, 0, 0]]),
              outliers=dict(n_plane=400, n_edge=40, noise=0.01),
              many_edges=dict(n_plane=500, n_edge=2600, noise=0.01))[case]
    corr, truth = _synthetic_corr(rng, **kw)
    if case == "outliers":
        corr["C"][::7] += rng.normal(scale=3.0, size=(len(corr[::7]), 3))
    if case == "many_edges":
        # more edges than the solver's dense edge list holds (1 024), 60 % of them rejected correspondences (kind 0,
        # N = C = 0) on both sides of that limit: the listed part and the checked tail must add up to the oracle's sum
        drop = rng.random(2600) < 0.6
        idx = np.nonzero(drop)[0]
        corr["kind"][idx] = 0
        corr["N"][idx] = 0.0
        corr["C"][idx] = 0.0
    guess = truth.copy() if case == "at_optimum" else synth.perturb_pose(truth, rng, 0.2, 2.0)
    ne = kw["n_edge"]
    corner = np.concatenate([corr["p"][:ne], np.zeros((ne, 1))], 1).astype(np.float32)
    surf = np.concatenate([corr["p"][ne:], np.zeros((len(corr) - ne, 1))], 1).astype(np.float32)
    rec = np.concatenate([corr["C"], corr["N"]], 1)
    pose_o, summ = oracle.ceres_solve(corr, guess)
    pose_g, info = gpu.solve_records(corner, surf, rec, guess)
    assert info.lm_iterations[0] == summ.iterations and info.lm_successful[0] == summ.successful_steps, case
    dt, dr = synth.pose_error(pose_g, pose_o)
    assert dt < 1e-6 and dr < 1e-6, (case, dt, dr)
    if case in ("generic", "edges_only", "outliers", "many_edges"):
        et, er = synth.pose_error(pose_g, truth)
        assert et < 0.05 and er < 0.01
    if case == "at_optimum":
        assert np.array_equal(pose_g, guess) or synth.pose_error(pose_g, guess)[0] < 1e-9


@pytest.mark.gpu
def test_candidate_counter_mode_counts_and_keeps_results(gpu):
    """msfl_set_timing(h, 3): the counting 5-NN instantiation returns the same poses and a plausible count."""
    import bench
    inp = bench.build_inputs(8, 20000, 0)
    h = gpu
    h.set_map(inp["map_corner"], inp["map_surf"])
    args = (inp["corner"], inp["corner_off"], inp["surf"], inp["surf_off"])
    p0, s0, _ = h.match_scan2map_batch(*args, inp["guesses"].copy())
    h.set_timing(3)
    h.get_timing(reset=True)
    p1, s1, _ = h.match_scan2map_batch(*args, inp["guesses"].copy())
    t = h.get_timing(reset=True)
    h.set_timing(0)
    assert np.array_equal(p0, p1) and np.array_equal(s0, s1)
    n_query = 2 * (len(inp["corner"]) + len(inp["surf"]))                # two outer iterations
    assert t.launches_assoc == 2
    n_cand = t.knn_candidates + t.knn_candidates_seeded               # first pass (from the gate) + second pass (seeded bound)
    assert 5 * n_query * 0.5 < n_cand < 2000 * n_query                # at least ~5 per accepted query, far below the map size
    assert 0 < t.knn_candidates_seeded                                # the second pass is counted on its own
    assert t.launches_assoc_seeded == 1 and 0 < t.ms_assoc_seeded < t.ms_assoc
    t0 = h.get_timing(reset=False)
    assert t0.knn_candidates == 0 and t0.knn_candidates_seeded == 0      # reset


def test_both_forms_of_the_5nn_search_agree_bit_for_bit(oracle, monkeypatch):
    """Launches of up to 32 768 queries take the row-parallel latency form of the 5-NN kernel (sixteen lanes per query, one
    lane per (y, z) row, five-round merge), larger ones the one-lane-per-query form.  An exact top-5 over (distance,
    index) keys does not depend on the visit order: records, poses and iteration counts must be equal bit for bit, on
    the lattice map (ties, duplicates, gate boundaries) and on scans."""
    from msf_loam_amd import capi
    hs = {}
    for form in ("lane", "rows"):
        monkeypatch.setenv("MSFL_KNN_FORM", form)
        hs[form] = capi.Handle(0)
    monkeypatch.delenv("MSFL_KNN_FORM")
    try:
        _, mc, ms = common.small_world()
        for h in hs.values():
            h.set_map(mc, ms)
        n_acc = 0
        for pts, ring, truth, guess in common.scans(4):
            _, corner, surf = common.features_from_oracle(oracle, pts, ring)
            rec = {f: h.associate_scan2map(corner, surf, guess) for f, h in hs.items()}
            assert np.array_equal(rec["lane"], rec["rows"])
            n_acc += int(np.any(rec["rows"][:, 3:] != 0, axis=1).sum())
            out = {f: h.match_scan2map(corner, surf, guess) for f, h in hs.items()}
            assert np.array_equal(out["lane"][1], out["rows"][1])
            assert list(out["lane"][2].lm_iterations) == list(out["rows"][2].lm_iterations)
        assert n_acc > 4000
        # sparse and empty neighbourhoods, queries outside the index's bounding box, a map of five points
        rng = np.random.default_rng(5)
        tiny = np.concatenate([rng.uniform(-2, 2, (5, 3)), np.zeros((5, 1))], 1).astype(np.float32)
        cloud = np.concatenate([rng.uniform(-30, 30, (3000, 3)), np.zeros((3000, 1))], 1).astype(np.float32)
        far = np.concatenate([rng.uniform(-80, 80, (500, 3)), np.zeros((500, 1))], 1).astype(np.float32)
        for m_c, m_s in ((tiny, cloud), (cloud[:700], tiny), (cloud[:40], cloud)):
            for h in hs.values():
                h.set_map(m_c, m_s)
            rec = {f: h.associate_scan2map(far[:100], far, np.array([0.3, -0.2, 0.1, 0, 0, 0, 1.0])) for f, h in hs.items()}
            assert np.array_equal(rec["lane"], rec["rows"])
    finally:
        for h in hs.values():
            h.close()


def _lattice_job(rng):
    """Lattice maps (exactly equal f32 distances, duplicated points) and queries on nodes, half and quarter pitches."""
    g_ = np.arange(-6, 7, dtype=np.float32) * 0.5
    X, Y = np.meshgrid(g_, g_, indexing="ij")
    plane = np.stack([X.ravel(), Y.ravel(), np.full(X.size, -1.5, np.float32)], 1)
    wall = np.stack([np.full(X.size, 3.5, np.float32), X.ravel(), Y.ravel() + 1.5], 1)
    lat = np.concatenate([plane, wall, plane[::7]])
    lat = np.concatenate([lat, np.zeros((len(lat), 1), np.float32)], 1).astype(np.float32)
    lat = lat[rng.permutation(len(lat))]
    line = np.stack([np.zeros(60, np.float32), np.zeros(60, np.float32), np.arange(60, dtype=np.float32) * 0.125], 1)
    pole = np.concatenate([line, line[::5]]); pole = np.concatenate([pole, np.zeros((len(pole), 1), np.float32)], 1).astype(np.float32)
    q = np.concatenate([plane[rng.integers(0, len(plane), 300)] + rng.choice([0.0, 0.25, 0.125], (300, 3)).astype(np.float32),
                        wall[rng.integers(0, len(wall), 300)] + rng.choice([0.0, 0.25, -0.125], (300, 3)).astype(np.float32),
                        np.array([[0, 0, -0.5], [0, 0, -0.5001], [0, 0, -0.54], [2.999, 0, -1.5], [0.5, 0.5, -0.5]], np.float32)])
    surf = np.concatenate([q, np.zeros((len(q), 1), np.float32)], 1).astype(np.float32)
    corner = np.concatenate([line[::3] + np.float32(0.05), np.zeros((20, 1), np.float32)], 1).astype(np.float32)
    return pole, lat, corner, surf


def test_truncated_key_walk_is_exact_on_ties_and_at_the_gate(oracle, monkeypatch):
    """Round 4: the one-lane-per-query kernels walk with 32-bit keys (distance truncated to 29 bits | slot) and search a query
    again with the exact (distance, index) keys when its final keys cannot decide the top-5 (two distances in one 2^-21 bucket,
    a 5th distance in the gate's bucket).  On lattice maps nearly every query is such a tie: the mixed kernel (`lane`), the
    per-kind kernel of batches of >= 65 536 features and the row-parallel form (exact keys only) must agree bit for bit, and
    the accept sets must be the oracle's."""
    from msf_loam_amd import capi
    hs = {}
    for form in ("lane", "rows"):
        monkeypatch.setenv("MSFL_KNN_FORM", form)
        hs[form] = capi.Handle(0)
    monkeypatch.delenv("MSFL_KNN_FORM")
    try:
        rng = np.random.default_rng(41)
        pole, lat, corner, surf = _lattice_job(rng)
        poses = [np.array([0, 0, 0, 0, 0, 0, 1.0]), np.array([0.125, -0.25, 0.0, 0, 0, 0, 1.0]), np.array([0.25, 0.25, 0.5, 0, 0, 0, 1.0])]
        for k in range(5):
            p = np.r_[rng.normal(0, 0.05, 3), 0.5 * rng.normal(0, 0.004, 3), 1.0]; p[3:] /= np.linalg.norm(p[3:]); poses.append(p)
        for h in hs.values():
            h.set_map(pole, lat)
        n_ok = 0
        for pose in poses:
            rec = {f: h.associate_scan2map(corner, surf, pose) for f, h in hs.items()}
            assert np.array_equal(rec["lane"], rec["rows"])
            corr = oracle.associate_scan2map(pole, lat, corner, surf, pose)
            assert np.array_equal(np.any(rec["lane"][:, 3:] != 0, axis=1), corr["kind"] != 0)
            n_ok += int((corr["kind"] != 0).sum())
        assert n_ok > 1500
        # the per-kind kernel: 120 copies of the job (75 000 features) in one batch against single calls through the rows form
        B = 120
        guesses = np.array([poses[i % len(poses)] for i in range(B)])
        co = np.arange(B + 1, dtype=np.int32) * len(corner); so = np.arange(B + 1, dtype=np.int32) * len(surf)
        assert co[-1] + so[-1] >= 65536
        h = hs["lane"]
        pb, sb, ib = h.match_scan2map_batch(np.tile(corner, (B, 1)), co, np.tile(surf, (B, 1)), so, guesses, want_info=True)
        for i in range(len(poses)):
            s1, p1, i1 = hs["rows"].match_scan2map(corner, surf, poses[i])
            for j in range(i, B, len(poses)):
                assert sb[j] == s1 and np.array_equal(pb[j], p1), (i, j)
                assert list(ib[j].n_plane) == list(i1.n_plane) and list(ib[j].n_edge) == list(i1.n_edge)
                assert list(ib[j].final_cost) == list(i1.final_cost)
    finally:
        for h in hs.values():
            h.close()


@pytest.mark.parametrize("form", ["lane", "split"])
def test_seeded_second_pass_finds_the_same_neighbours(oracle, monkeypatch, form):
    """The second outer iteration's 5-NN search starts from the bound the first iteration's five neighbours give
    (knn5_seed_bound) instead of the acceptance gate.  It must remain the EXACT top-5 by (distance, index): against a handle
    with MSFL_KNN_SEED=0 the poses, statuses, accepted counts, iteration counts and costs are equal bit for bit -- on
    thinned maps with holes and guesses up to a metre off (neighbour sets that change between the passes, features accepted
    in one pass only), on the lattice map (ties at the bound, duplicates), and through both one-lane-per-query kernels
    (`lane`: the mixed kernel of small batches; `split`: the per-kind kernel of batches of >= 65 536 features)."""
    from msf_loam_amd import capi
    monkeypatch.setenv("MSFL_KNN_FORM", "lane")
    monkeypatch.setenv("MSFL_KNN_SEED", "0")
    h0 = capi.Handle(0)
    monkeypatch.setenv("MSFL_KNN_SEED", "1")
    h1 = capi.Handle(0)
    try:
        _, mc, ms = common.small_world()
        n_seeds = max(int(os.environ.get("MSFL_FUZZ_SEEDS", "6")), 6)
        reps = 16 if form == "split" else 1          # 16 x ~5 000 features per job: beyond the split kernel's 65 536-feature threshold
        for seed in range(n_seeds):
            rng = np.random.default_rng(9100 + seed)
            keep_c = rng.uniform(size=len(mc)) < rng.uniform(0.3, 1.0)
            keep_s = rng.uniform(size=len(ms)) < rng.uniform(0.15, 1.0)
            lo = rng.uniform(-15, 5, 3); hi = lo + rng.uniform(2, 12, 3)
            keep_s &= ~np.all((ms[:, :3] > lo) & (ms[:, :3] < hi), axis=1)
            mc2, ms2 = np.ascontiguousarray(mc[keep_c]), np.ascontiguousarray(ms[keep_s])
            cs, ss, co, so, gs = [], [], [0], [0], []
            for r in range(reps):
                for pts, ring, truth, guess in common.scans(4):
                    _, corner, surf = common.features_from_oracle(oracle, pts, ring)
                    g = np.array(truth, np.float64)
                    g[:3] += rng.normal(0, rng.choice([0.05, 0.3, 0.6]), 3)
                    g[3:] = synth.quat_mul(g[3:], np.r_[0.5 * rng.normal(0, rng.choice([0.005, 0.03]), 3), 1.0]); g[3:] /= np.linalg.norm(g[3:])
                    cs.append(corner); ss.append(surf); co.append(co[-1] + len(corner)); so.append(so[-1] + len(surf)); gs.append(g)
            args = (np.concatenate(cs), np.array(co, np.int32), np.concatenate(ss), np.array(so, np.int32))
            if form == "split":
                assert co[-1] + so[-1] >= 65536
            out = []
            for h in (h0, h1):
                h.set_map(mc2, ms2)
                out.append(h.match_scan2map_batch(*args, np.array(gs), want_info=True))
            (p0, s0, i0), (p1, s1, i1) = out
            assert np.array_equal(p0, p1) and np.array_equal(s0, s1)
            for a, b in zip(i0, i1):
                assert list(a.n_edge) == list(b.n_edge) and list(a.n_plane) == list(b.n_plane) and list(a.lm_iterations) == list(b.lm_iterations)
                assert list(a.final_cost) == list(b.final_cost)
            if seed == 0:                              # the second pass really runs seeded, on fewer candidates
                h1.set_timing(3); h1.get_timing(reset=True)
                h1.match_scan2map_batch(*args, np.array(gs))
                t = h1.get_timing(reset=True); h1.set_timing(0)
                assert 0 < t.knn_candidates_seeded < 0.95 * t.knn_candidates
        if form == "lane":
            # lattice map: exactly equal f32 distances at the bound, duplicated points; two different small motions between the passes
            g_ = np.arange(-6, 7, dtype=np.float32) * 0.5
            X, Y = np.meshgrid(g_, g_, indexing="ij")
            plane = np.stack([X.ravel(), Y.ravel(), np.full(X.size, -1.5, np.float32)], 1)
            wall = np.stack([np.full(X.size, 3.5, np.float32), X.ravel(), Y.ravel() + 1.5], 1)
            lat = np.concatenate([plane, wall, plane[::7]])
            lat = np.concatenate([lat, np.zeros((len(lat), 1), np.float32)], 1).astype(np.float32)
            line = np.stack([np.zeros(60, np.float32), np.zeros(60, np.float32), np.arange(60, dtype=np.float32) * 0.125], 1)
            pole = np.concatenate([line, line[::5]]); pole = np.concatenate([pole, np.zeros((len(pole), 1), np.float32)], 1).astype(np.float32)
            rng = np.random.default_rng(4)
            q = np.concatenate([plane[rng.integers(0, len(plane), 300)] + rng.choice([0.0, 0.25, 0.125], (300, 3)).astype(np.float32),
                                wall[rng.integers(0, len(wall), 300)] + rng.choice([0.0, 0.25, -0.125], (300, 3)).astype(np.float32)])
            surf = np.concatenate([q, np.zeros((len(q), 1), np.float32)], 1).astype(np.float32)
            corner = np.concatenate([line[::3] + np.float32(0.05), np.zeros((20, 1), np.float32)], 1).astype(np.float32)
            for pose in (np.array([0, 0, 0, 0, 0, 0, 1.0]), np.array([0.125, -0.25, 0.0, 0, 0, 0, 1.0]), np.array([0.01, 0.02, -0.03, 0, 0, 0.002, 1.0])):
                res = []
                for h in (h0, h1):
                    h.set_map(pole, lat)
                    res.append(h.match_scan2map(corner, surf, pose / np.r_[1, 1, 1, [np.linalg.norm(pose[3:])] * 4]))
                assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1])
                assert list(res[0][2].n_plane) == list(res[1][2].n_plane) and list(res[0][2].final_cost) == list(res[1][2].final_cost)
    finally:
        h0.close(); h1.close()


@pytest.mark.parametrize("kind", ["outdoor", "corridor"])
def test_other_worlds_follow_the_oracle(gpu, oracle, kind):
    """Round 5: the registration off its home field.  `outdoor`: 680 k map points, a quarter of the occupied
    cells leaf-dense volumes (hundreds of candidates in a query's 27 cells; five neighbours that fit no line / plane and are
    rejected by the eigenvalue ratio / the 0.2 m plane test); `corridor`: the along-axis direction nearly unobservable, trust-region
    steps rejected.  Accept sets, accepted counts and LM iteration / success counts equal the oracle's, records <= 1e-9, poses <= 1e-7;
    the batch call and both 5-NN forms give the single call's bits (mapping_scan_matcher.cc:109-259)."""
    from msf_loam_amd import capi
    _, mc, ms = common.other_world(kind)
    gpu.set_map(mc, ms)
    cs, ss, co, so, guesses, singles = [], [], [0], [0], [], []
    n_rejected_fits, n_rejected_steps = 0, 0
    for pts, ring, truth, guess in common.other_scans(kind, 4):
        _, corner, surf = common.features_from_oracle(oracle, pts, ring)
        rec_g = gpu.associate_scan2map(corner, surf, guess)
        rec_o, corr = _oracle_records(oracle, mc, ms, corner, surf, guess)
        acc_o = corr["kind"] != 0
        assert np.array_equal(np.any(rec_g[:, 3:] != 0, axis=1), acc_o), "accepted-correspondence sets differ"
        n_rejected_fits += int((~acc_o).sum())
        n_dot = np.abs(np.sum(rec_g[acc_o, 3:] * rec_o[acc_o, 3:], axis=1))
        assert np.all(np.abs(n_dot - 1) < 1e-9)
        pl = acc_o.copy(); pl[:len(corner)] = False
        assert np.abs(rec_g[pl, :3] - rec_o[pl, :3]).max() < 1e-9
        rc, pose_o, info_o = oracle.match_scan2map(mc, ms, corner, surf, guess)
        s, pose_g, info_g = gpu.match_scan2map(corner, surf, guess)
        assert s == 0 and rc == 0
        assert list(info_g.n_edge) == list(info_o.n_edge) and list(info_g.n_plane) == list(info_o.n_plane)
        assert list(info_g.lm_iterations) == list(info_o.lm_iterations) and list(info_g.lm_successful) == list(info_o.lm_successful)
        n_rejected_steps += sum(info_o.lm_iterations) - sum(info_o.lm_successful)
        dt, dr = synth.pose_error(pose_g, pose_o)
        assert dt < TIGHT and dr < TIGHT, (kind, dt, dr)
        cs.append(corner); ss.append(surf); co.append(co[-1] + len(corner)); so.append(so[-1] + len(surf)); guesses.append(guess); singles.append(pose_g)
    assert n_rejected_fits > 500                       # the worlds do produce neighbourhoods that are neither line nor plane
    if kind == "corridor":
        assert n_rejected_steps > 0
    poses, st, _ = gpu.match_scan2map_batch(np.concatenate(cs), co, np.concatenate(ss), so, guesses)
    assert np.all(st == 0) and all(np.array_equal(poses[b], singles[b]) for b in range(4))
    # 16 copies of the batch go through the per-kind kernel of batches of >= 65 536 features (outdoor only reaches that)
    reps = 16
    C, S = np.concatenate(cs * reps), np.concatenate(ss * reps)
    co2 = np.cumsum([0] + [len(c) for c in cs * reps]).astype(np.int32); so2 = np.cumsum([0] + [len(x) for x in ss * reps]).astype(np.int32)
    poses2, st2, _ = gpu.match_scan2map_batch(C, co2, S, so2, guesses * reps)
    assert np.all(st2 == 0) and all(np.array_equal(poses2[b], singles[b % 4]) for b in range(4 * reps))
    if kind == "outdoor":
        assert co2[-1] + so2[-1] >= 65536
    # the row-parallel latency form of the search on the same map
    import os as _os
    old = _os.environ.get("MSFL_KNN_FORM")
    _os.environ["MSFL_KNN_FORM"] = "rows"
    try:
        h = capi.Handle(0)
    finally:
        if old is None:
            del _os.environ["MSFL_KNN_FORM"]
        else:
            _os.environ["MSFL_KNN_FORM"] = old
    try:
        h.set_map(mc, ms)
        for b in range(2):
            s, p, _ = h.match_scan2map(cs[b], ss[b], guesses[b])
            assert s == 0 and np.array_equal(p, singles[b])
    finally:
        h.close()


def test_deferred_pivoted_qr_planes_equal_the_inline_fallback(gpu, oracle):
    """Round 5: the whole-batch fit kernel (batches of >= 65 536 features) no longer carries the pivoted-QR fallback of the plane fit; a
    neighbourhood that fails the guards of the adjugate form (planes through the origin: A n = -1 has no solution; collinear
    neighbours; a plane seen from 1 km away) goes to a list and is fitted by fit_fallback_kernel.  120 copies of a job that is FULL of
    such neighbourhoods, against the single-call path (which keeps the fallback inline) bit for bit, and the oracle's accept sets."""
    rng = np.random.default_rng(61)
    g_ = np.arange(-6, 7, dtype=np.float32) * 0.5
    X, Y = np.meshgrid(g_, g_, indexing="ij")
    origin_plane = np.stack([X.ravel(), Y.ravel(), np.zeros(X.size, np.float32)], 1)                  # z = 0: through the origin
    tilted = np.stack([X.ravel() + 20, Y.ravel(), 0.3 * (X.ravel() + 20) - 0.2 * Y.ravel()], 1)         # another plane through the origin
    far = np.stack([X.ravel() + 1000.0, Y.ravel() + 1000.0, np.full(X.size, 2.0, np.float32)], 1)       # |c| >> spread: the grazing guard
    line = np.stack([np.arange(40, dtype=np.float32) * 0.25 - 5, np.full(40, 40.0, np.float32), np.full(40, 1.0, np.float32)], 1)   # collinear: rank 1 after centring
    ok_plane = np.stack([X.ravel() - 20, Y.ravel(), np.full(X.size, -1.5, np.float32)], 1)
    ms = np.concatenate([origin_plane, tilted, far, line, ok_plane]).astype(np.float32)
    ms = np.concatenate([ms + rng.normal(0, 1e-4, ms.shape).astype(np.float32) * (np.arange(len(ms)) % 3 == 0)[:, None], np.zeros((len(ms), 1), np.float32)], 1).astype(np.float32)
    ms = ms[rng.permutation(len(ms))]
    pole = np.stack([np.zeros(60, np.float32), np.zeros(60, np.float32) + 60, np.arange(60, dtype=np.float32) * 0.125], 1)
    mc = np.concatenate([pole, np.zeros((60, 1), np.float32)], 1).astype(np.float32)
    def around(p, n):
        return p[rng.integers(0, len(p), n)] + rng.uniform(-0.2, 0.2, (n, 3)).astype(np.float32)
    q = np.concatenate([around(origin_plane, 150), around(tilted, 150), around(far, 120), around(line, 60), around(ok_plane, 120)]).astype(np.float32)
    surf = np.concatenate([q, np.zeros((len(q), 1), np.float32)], 1).astype(np.float32)
    corner = np.concatenate([pole[::3] + np.float32(0.05), np.zeros((20, 1), np.float32)], 1).astype(np.float32)
    gpu.set_map(mc, ms)
    poses = [np.array([0, 0, 0, 0, 0, 0, 1.0]), np.array([0.05, -0.02, 0.01, 0, 0, 0.001, 1.0]), np.array([-0.03, 0.04, 0.0, 0.001, 0, 0, 1.0])]
    poses = [p / np.r_[1, 1, 1, [np.linalg.norm(p[3:])] * 4] for p in poses]
    n_acc = 0
    for pose in poses:
        rec = gpu.associate_scan2map(corner, surf, pose)
        corr = oracle.associate_scan2map(mc, ms, corner, surf, pose)
        assert np.array_equal(np.any(rec[:, 3:] != 0, axis=1), corr["kind"] != 0)
        n_acc += int((corr["kind"] != 0).sum())
    assert n_acc > 300
    B = 120
    guesses = np.array([poses[i % 3] for i in range(B)])
    co = np.arange(B + 1, dtype=np.int32) * len(corner); so = np.arange(B + 1, dtype=np.int32) * len(surf)
    assert co[-1] + so[-1] >= 65536
    for rep in range(3):                             # the two fallback lists alternate: three batches use both and re-arm both
        pb, sb, ib = gpu.match_scan2map_batch(np.tile(corner, (B, 1)), co, np.tile(surf, (B, 1)), so, guesses, want_info=True)
        for i in range(3):
            s1, p1, i1 = gpu.match_scan2map(corner, surf, poses[i])
            for j in range(i, B, 3):
                assert sb[j] == s1 and np.array_equal(pb[j], p1), (rep, i, j)
                assert list(ib[j].n_plane) == list(i1.n_plane) and list(ib[j].final_cost) == list(i1.final_cost)


def test_pair_index_build_equals_two_single_builds(oracle, monkeypatch):
    """Round 6.  msfl_set_map indexes both maps through ONE chain of launches (bounding boxes, counts, one prefix sum over both
    cell tables, scatter); MSFL_INDEX_SINGLE=1 keeps the two single builds of rounds 1-5.  The order of points inside a cell is
    arbitrary in both and never observable (the 5-NN orders by (distance, original index)): records, poses, iteration counts
    and costs must be equal bit for bit -- on scans, after maps of changing size and extent (the table spans adapt from
    build to build), with non-finite map points, on a five-point corner map, and with the stream's cached offset table."""
    from msf_loam_amd import capi
    hs = {"pair": capi.Handle(0)}
    monkeypatch.setenv("MSFL_INDEX_SINGLE", "1"); hs["single"] = capi.Handle(0); monkeypatch.delenv("MSFL_INDEX_SINGLE")
    try:
        _, mc, ms = common.small_world()
        rng = np.random.default_rng(8)
        scans = [(common.features_from_oracle(oracle, pts, ring)[1:], guess) for pts, ring, truth, guess in common.scans(4)]
        far = ms.copy(); far[:50, 0] += 400.0                                   # a much larger bounding box: the tables grow, then shrink again
        holes = ms.copy(); holes[rng.choice(len(ms), 40, replace=False), rng.integers(0, 3, 40)] = np.nan
        tiny_c = mc[:5].copy()
        maps = [(mc, ms), (mc[:len(mc) // 2], ms[::3]), (mc, far), (mc, ms), (tiny_c, holes), (mc, ms)]
        for k, (a, b) in enumerate(maps):
            for h in hs.values():
                h.set_map(a, b)
            for (corner, surf), guess in scans[:2] if k else scans:
                rec = {f: h.associate_scan2map(corner, surf, guess) for f, h in hs.items()}
                assert np.array_equal(rec["pair"], rec["single"]), k
                out = {f: h.match_scan2map(corner, surf, guess) for f, h in hs.items()}
                assert out["pair"][0] == out["single"][0] and np.array_equal(out["pair"][1], out["single"][1]), k
                for f in ("lm_iterations", "n_edge", "n_plane", "final_cost"):
                    assert list(getattr(out["pair"][2], f)) == list(getattr(out["single"][2], f)), (k, f)
        # a batch registered twice with a map rebuild in between (same offsets: the second call skips the offset upload) and once with
        # different offsets in between
        cs = [s[0][0] for s in scans]; ss = [s[0][1] for s in scans]; gs = [s[1] for s in scans]
        co = np.cumsum([0] + [len(c) for c in cs]); so = np.cumsum([0] + [len(c) for c in ss])
        C, S = np.concatenate(cs), np.concatenate(ss)
        for h in hs.values():
            h.set_map(mc, ms)
        first = {f: h.match_scan2map_batch(C, co, S, so, gs) for f, h in hs.items()}
        assert np.array_equal(first["pair"][0], first["single"][0]) and np.all(first["pair"][1] == 0)
        for h in hs.values():
            h.set_map(mc, ms)
        again = hs["pair"].match_scan2map_batch(C, co, S, so, gs)
        assert np.array_equal(again[0], first["pair"][0])
        part = hs["pair"].match_scan2map_batch(C[:co[2]], co[:3], S[:so[2]], so[:3], gs[:2])
        assert np.array_equal(part[0], first["pair"][0][:2])
        again = hs["pair"].match_scan2map_batch(C, co, S, so, gs)
        assert np.array_equal(again[0], first["pair"][0])
        # one empty map: the single builds serve it (the pair chain needs both clouds), registrations fail with NO_MAP-style statuses alike
        for h in hs.values():
            h.set_map(mc, ms[:0])
        st = {f: h.match_scan2map(cs[0], ss[0], gs[0], allow=(capi.NO_MAP, capi.BAD_ARG, capi.MAP_TOO_SMALL))[0] for f, h in hs.items()}
        assert st["pair"] == st["single"] != 0
    finally:
        for h in hs.values():
            h.close()


def test_the_batch_step_replays_from_a_captured_graph(oracle):
    """Round 6.  Once its buffers exist, msfl_set_map + msfl_match_scan2map_batch on device-resident inputs enqueue kernels and one memset
    only (no copy, event, allocation or synchronisation: the pair index build reports its wanted table size through pinned host memory,
    an unchanged offset table is not uploaded again), so the whole step can be captured into a HIP graph on the caller's stream and
    replayed.  Replays must reproduce the eager poses bit for bit -- also after the poses were overwritten and the guesses changed."""
    torch = pytest.importorskip("torch")
    from msf_loam_amd import capi
    _, mc, ms = common.small_world()
    cs, ss, co, so, guesses = [], [], [0], [0], []
    for pts, ring, truth, guess in common.scans(6):
        _, corner, surf = common.features_from_oracle(oracle, pts, ring)
        cs.append(corner); ss.append(surf); co.append(co[-1] + len(corner)); so.append(so[-1] + len(surf)); guesses.append(guess)
    co = np.array(co, np.int32); so = np.array(so, np.int32)
    dev = torch.device("cuda", 0)
    h = capi.Handle(0)
    try:
        s = torch.cuda.Stream(dev)
        h.set_stream(s.cuda_stream)
        with torch.cuda.stream(s):
            d_mc = torch.from_numpy(mc).to(dev); d_ms = torch.from_numpy(ms).to(dev)
            d_c = torch.from_numpy(np.concatenate(cs)).to(dev); d_s = torch.from_numpy(np.concatenate(ss)).to(dev)
            d_guess = torch.from_numpy(np.array(guesses)).to(dev)
            d_poses = torch.zeros((len(guesses), 7), dtype=torch.float64, device=dev); d_status = torch.zeros(len(guesses), dtype=torch.int32, device=dev)

            def step():
                d_poses.copy_(d_guess)
                h.set_map(d_mc, d_ms, len(mc), len(ms), capi.MEM_DEVICE)
                h.match_scan2map_batch_device(len(guesses), d_c, co, d_s, so, d_poses, d_status)
            for _ in range(3):
                step()
            torch.cuda.synchronize(dev)
            eager = d_poses.cpu().numpy().copy()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            step()
        for _ in range(3):
            d_poses.zero_()
            g.replay()
            torch.cuda.synchronize(dev)
            assert np.array_equal(d_poses.cpu().numpy(), eager)
        assert (d_status.cpu().numpy() == 0).all()
        # other guesses through the same graph (the graph reads the guess buffer, not its contents at capture time)
        with torch.cuda.stream(s):
            d_guess.copy_(torch.from_numpy(np.array(guesses[::-1]).copy()).to(dev))
            torch.cuda.synchronize(dev)
        g.replay(); torch.cuda.synchronize(dev)
        swapped = d_poses.cpu().numpy().copy()
        with torch.cuda.stream(s):
            step(); torch.cuda.synchronize(dev)
        assert np.array_equal(d_poses.cpu().numpy(), swapped)
        _, p0, _ = h.match_scan2map(cs[0], ss[0], guesses[-1])
        assert np.array_equal(p0, swapped[0])
    finally:
        h.close()
