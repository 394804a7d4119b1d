"""GPU parity of stage B (scan-to-scan registration) against the CPU oracle through the C ABI."""
import os

import numpy as np
import pytest

from msf_loam_amd import capi, synth
from tests import common

pytestmark = pytest.mark.gpu
TIGHT = 1e-7


def _pair(oracle, i):
    """Two consecutive scans of a slow trajectory + the oracle's features for both."""
    w, _, _ = common.small_world()
    base = synth.random_poses(8, synth.SEED + 77)[i]
    rng = np.random.default_rng(100 + i)
    nxt = synth.perturb_pose(base, rng, 0.25, 2.0)
    pa, ra = synth.make_scan(w, base, 9000 + i)
    pb, rb = synth.make_scan(w, nxt, 9100 + i)
    fa, fb = oracle.extract_features(pa, ra), oracle.extract_features(pb, rb)
    return fa, fb


def _clouds(fa, fb):
    return (fa["full"][fa["less_sharp"]], fa["ring"][fa["less_sharp"]], fa["full"][fa["less_flat"]], fa["ring"][fa["less_flat"]],
            fb["full"][fb["sharp"]], fb["full"][fb["flat"]])


def test_pose_parity_and_counts(gpu, oracle):
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    for i in range(4):
        fa, fb = _pair(oracle, i)
        c = _clouds(fa, fb)
        rc, pose_o, info_o = oracle.match_scan2scan(*c, ident)
        s, pose_g, info_g = gpu.match_scan2scan(*c, ident)
        assert s == rc == 0
        assert list(info_g.n_edge) == list(info_o.n_edge) and list(info_g.n_plane) == list(info_o.n_plane)
        assert list(info_g.lm_iterations) == list(info_o.lm_iterations)
        dt, dr = synth.pose_error(pose_g, pose_o)
        assert dt < 1e-4 and dr < 1e-4
        assert dt < TIGHT and dr < TIGHT, (dt, dr)
        assert info_o.n_edge[0] > 20 and info_o.n_plane[0] > 100


def test_batch_and_too_few_correspondences(gpu, oracle):
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    pairs = [_clouds(*_pair(oracle, i)) for i in range(3)]
    # pair 1 gets a hopeless guess: 1-NN beyond 5 m everywhere -> < 10 correspondences -> false (:262-267)
    guesses = np.stack([ident, np.array([500.0, 0, 0, 0, 0, 0, 1.0]), ident])
    sets = []
    for k in range(4):
        pts = np.concatenate([p[k if k < 2 else k + 2] if False else (p[0], p[2], p[4], p[5])[k] for p in pairs])
        off = np.cumsum([0] + [len((p[0], p[2], p[4], p[5])[k]) for p in pairs]).astype(np.int32)
        ring = np.concatenate([(p[1], p[3])[k] for p in pairs]) if k < 2 else None
        sets.append((pts, ring, off))
    poses, status, info = gpu.match_scan2scan_batch(sets, guesses, want_info=True)
    assert list(status) == [0, capi.TOO_FEW_CORRESPONDENCES, 0]
    assert np.array_equal(poses[1], guesses[1]), "a failed scan keeps its pose"
    for b in (0, 2):
        rc, pose_o, _ = oracle.match_scan2scan(*pairs[b], guesses[b])
        assert rc == 0
        dt, dr = synth.pose_error(poses[b], pose_o)
        assert dt < TIGHT and dr < TIGHT
    rc, pose_o, _ = oracle.match_scan2scan(*pairs[1], guesses[1])
    assert rc == 1 and np.array_equal(pose_o, guesses[1])
    s, p, _ = gpu.match_scan2scan(*pairs[1], guesses[1])
    assert s == capi.TOO_FEW_CORRESPONDENCES


def test_unsorted_rings_follow_reference_break_semantics(gpu, oracle):
    """The window scans `break` at the first out-of-window ring; on a cloud that is NOT ring-sorted
    that differs from a ring filter.  Shuffle the previous scan's feature order and compare."""
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    fa, fb = _pair(oracle, 5)
    c = list(_clouds(fa, fb))
    rng = np.random.default_rng(5)
    for k in (0, 2):
        perm = rng.permutation(len(c[k]))
        blocks = np.array_split(perm, 40)              # shuffle blocks: locally ordered, globally not
        perm = np.concatenate([np.sort(b) for b in blocks])
        c[k], c[k + 1] = c[k][perm], c[k + 1][perm]
    rc, pose_o, info_o = oracle.match_scan2scan(*c, ident)
    s, pose_g, info_g = gpu.match_scan2scan(*c, ident)
    assert s == rc
    assert list(info_g.n_edge) == list(info_o.n_edge) and list(info_g.n_plane) == list(info_o.n_plane)
    dt, dr = synth.pose_error(pose_g, pose_o)
    assert dt < TIGHT and dr < TIGHT


def test_tiled_throughput_kernel_equals_wave_latency_kernel(gpu, oracle, monkeypatch):
    """Small calls against a small previous cloud use one wavefront per query (a single pair with a full-size cloud goes
    through the column grid like a batch), large batches the LDS-tiled kernel / the column grid.  A 42-pair batch must
    reproduce the single-pair calls bit for bit, whichever kernel those take."""
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    base = [_clouds(*_pair(oracle, i)) for i in range(3)]
    pairs = [base[i % 3] for i in range(42)]
    guesses = np.stack([ident] * 42)
    guesses[:, 0] = np.linspace(-0.05, 0.05, 42)              # distinct guesses -> distinct results
    sets = []
    for k in range(4):
        pick = lambda p: (p[0], p[2], p[4], p[5])[k]
        pts = np.concatenate([pick(p) for p in pairs])
        off = np.cumsum([0] + [len(pick(p)) for p in pairs]).astype(np.int32)
        ring = np.concatenate([(p[1], p[3])[k] for p in pairs]) if k < 2 else None
        sets.append((pts, ring, off))
    assert 42 * max(len(p[4]) + len(p[5]) for p in pairs) > 16384       # really on the tiled path
    poses, status, _ = gpu.match_scan2scan_batch(sets, guesses)
    assert np.all(status == 0)
    for b in (0, 1, 2, 20, 41):
        s, p, _ = gpu.match_scan2scan(*pairs[b], guesses[b])
        assert s == 0 and np.array_equal(p, poses[b]), b
    monkeypatch.setenv("MSFL_ODOM_WAVE_MAX_TARGETS", "1000000000")       # single pairs on the wave kernel whatever their size
    hw = capi.Handle(0)
    try:
        for b in (0, 1, 2, 20, 41):
            s, p, _ = hw.match_scan2scan(*pairs[b], guesses[b])
            assert s == 0 and np.array_equal(p, poses[b]), b
    finally:
        hw.close()
    rc, pose_o, _ = oracle.match_scan2scan(*pairs[7], guesses[7])
    assert max(synth.pose_error(poses[7], pose_o)) < TIGHT


def _batch_sets(pairs):
    sets = []
    for k in range(4):
        pick = lambda p: (p[0], p[2], p[4], p[5])[k]
        pts = np.concatenate([pick(p) for p in pairs])
        off = np.cumsum([0] + [len(pick(p)) for p in pairs]).astype(np.int32)
        ring = np.concatenate([(p[1], p[3])[k] for p in pairs]) if k < 2 else None
        sets.append((pts, ring, off))
    return sets


def test_column_grid_planes_equal_brute_force(gpu, oracle, monkeypatch):
    """Large batches look plane correspondences up in a per-pair column grid; pairs the grid cannot
    take (rings not sorted, a point beyond +-512 m) stay on the brute-force kernel inside the same
    launch.  Every pair must reproduce its single-pair (wave kernel) result bit for bit, also with
    guesses far enough off that many queries walk the widest neighbourhood or find nothing."""
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    base = [list(_clouds(*_pair(oracle, i))) for i in range(3)]
    rng = np.random.default_rng(11)
    shuffled = [np.copy(a) for a in base[1]]
    perm = np.concatenate([np.sort(b) for b in np.array_split(rng.permutation(len(shuffled[2])), 30)])
    shuffled[2], shuffled[3] = shuffled[2][perm], shuffled[3][perm]
    far = [np.copy(a) for a in base[2]]
    far[2] = np.concatenate([far[2], np.array([[600.0, 3.0, 1.0, 0.0]], np.float32)])
    far[3] = np.concatenate([far[3], far[3][-1:]])
    variants = [base[0], base[1], base[2], shuffled, far]
    pairs = [variants[i % 5] for i in range(45)]
    guesses = np.stack([ident] * 45)
    guesses[:, 0] = np.linspace(-2.5, 2.5, 45)               # up to 2.5 m off: sparse matches, wide walks
    guesses[:, 1] = np.linspace(1.0, -1.0, 45)
    guesses[44] = np.array([300.0, 0, 0, 0, 0, 0, 1.0])      # nothing within 5 m
    sets = _batch_sets(pairs)
    poses, status, info = gpu.match_scan2scan_batch(sets, guesses, want_info=True)
    assert status[44] == capi.TOO_FEW_CORRESPONDENCES
    for b in range(45):
        s, p, i1 = gpu.match_scan2scan(*pairs[b], guesses[b])
        assert s == status[b], b
        assert np.array_equal(p, poses[b]), b
        assert i1.n_plane[0] == info[b].n_plane[0] and i1.n_plane[1] == info[b].n_plane[1], b
    for b in (0, 3, 4, 22):
        rc, pose_o, _ = oracle.match_scan2scan(*pairs[b], guesses[b])
        assert rc == status[b]
        assert max(synth.pose_error(poses[b], pose_o)) < TIGHT
    # the same batch with the grid disabled
    monkeypatch.setenv("MSFL_ODOM_BRUTE", "1")
    h2 = capi.Handle(0)
    try:
        poses2, status2, _ = h2.match_scan2scan_batch(sets, guesses)
    finally:
        h2.close()
    assert np.array_equal(status, status2) and np.array_equal(poses, poses2)


def test_binning_a_cloud_longer_than_its_register_resident_part(gpu, oracle, monkeypatch):
    """The column-grid binning kernel keeps the first 20 x 1 024 points of a previous-scan cloud in registers between
    its passes and streams the rest.  A less-flat cloud of ~2.2x that length (ring-monotone, < 65 536 points: still
    the grid path) must give the brute-force kernel's poses bit for bit, and the oracle's within the parity bound."""
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    rng = np.random.default_rng(23)
    pairs = []
    for i in range(2):
        c = [np.copy(a) for a in _clouds(*_pair(oracle, i))]
        pts = [c[2]]
        for _ in range(2):                                        # two jittered copies: a denser previous scan
            q = np.copy(c[2]); q[:, :3] += rng.normal(0, 0.03, (len(q), 3)).astype(np.float32); pts.append(q)
        cat, ring = np.concatenate(pts)[: 45000], np.concatenate([c[3]] * 3)[: 45000]
        order = np.argsort(ring, kind="stable")
        c[2], c[3] = np.ascontiguousarray(cat[order]), np.ascontiguousarray(ring[order])
        assert 2 * 20 * 1024 < len(c[2]) < 65536
        pairs.append(c)
    batch = [pairs[i % 2] for i in range(40)]                     # a batch large enough for the throughput path
    guesses = np.stack([ident] * 40)
    guesses[:, 0] = np.linspace(-0.4, 0.4, 40)
    sets = _batch_sets(batch)
    poses, status, info = gpu.match_scan2scan_batch(sets, guesses, want_info=True)
    assert np.all(status == 0)
    monkeypatch.setenv("MSFL_ODOM_BRUTE", "1")
    h2 = capi.Handle(0)
    try:
        poses2, status2, _ = h2.match_scan2scan_batch(sets, guesses)
    finally:
        h2.close()
    assert np.array_equal(status, status2) and np.array_equal(poses, poses2)
    for b in (0, 1, 39):
        rc, pose_o, info_o = oracle.match_scan2scan(*batch[b], guesses[b])
        assert rc == 0 and list(info[b].n_plane) == list(info_o.n_plane)
        assert max(synth.pose_error(poses[b], pose_o)) < TIGHT


def test_binning_a_64_beam_sized_cloud_and_a_column_that_overflows_its_counter(gpu, oracle, monkeypatch):
    """A 64-beam less-flat list holds ~100 k points.  The binning kernel's u16 counters are per COLUMN, so such a cloud
    still takes the column grid (until round 5 every cloud above 65 535 points fell to the brute-force kernel: 27 ms per
    association inside the SLAM step); a cloud that really puts more than 65 535 points into one 1 m column is found by
    the histogram total and keeps the brute-force kernel.  Both must equal the brute-force poses bit for bit, the
    oracle's within the parity bound.  (What the grid buys shows in the 64-beam SLAM step, profiles/r05_slam64_*: the
    batch call here is dominated by its 72 MB of host-to-device copies.)"""
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    rng = np.random.default_rng(29)
    pairs = []
    for i in range(2):
        c = [np.copy(a) for a in _clouds(*_pair(oracle, i))]
        reps = -(-100000 // len(c[2]))
        pts = [c[2]]
        for _ in range(reps - 1):
            q = np.copy(c[2]); q[:, :3] += rng.normal(0, 0.03, (len(q), 3)).astype(np.float32); pts.append(q)
        cat, ring = np.concatenate(pts)[:100000], np.concatenate([c[3]] * reps)[:100000]
        if i == 1:                                                # 70 000 of them inside one column: the counter wraps
            cat[:70000, :2] = (np.array([3.5, 2.5]) + rng.uniform(-0.45, 0.45, (70000, 2))).astype(np.float32)
        order = np.argsort(ring, kind="stable")
        c[2], c[3] = np.ascontiguousarray(cat[order]), np.ascontiguousarray(ring[order])
        assert len(c[2]) == 100000
        pairs.append(c)
    batch = [pairs[0]] * 38 + [pairs[1]] * 2
    guesses = np.stack([ident] * 40)
    guesses[:, 0] = np.linspace(-0.4, 0.4, 40)
    sets = _batch_sets(batch)
    poses, status, info = gpu.match_scan2scan_batch(sets, guesses, want_info=True)
    assert np.all(status[:38] == 0)
    monkeypatch.setenv("MSFL_ODOM_BRUTE", "1")
    h2 = capi.Handle(0)
    try:
        poses2, status2, _ = h2.match_scan2scan_batch(sets, guesses)
    finally:
        h2.close()
    assert np.array_equal(status, status2) and np.array_equal(poses, poses2)
    # the index build with 64 workgroups per cloud (what a call with one to four such pairs takes: the SLAM step)
    monkeypatch.delenv("MSFL_ODOM_BRUTE")
    monkeypatch.setenv("MSFL_ODOM_BIN_SPLIT", "1")
    h3 = capi.Handle(0)
    try:
        poses3, status3, _ = h3.match_scan2scan_batch(sets, guesses)
        few = _batch_sets(batch[37:40])                           # three pairs: the split by itself (MSFL_ODOM_BIN_SPLIT unset below)
        monkeypatch.delenv("MSFL_ODOM_BIN_SPLIT")
        h4 = capi.Handle(0)
        try:
            poses4, status4, _ = h4.match_scan2scan_batch(few, guesses[37:40])
        finally:
            h4.close()
    finally:
        h3.close()
    assert np.array_equal(status, status3) and np.array_equal(poses, poses3)
    assert np.array_equal(status[37:40], status4) and np.array_equal(poses[37:40], poses4)
    for b in (0, 39):
        rc, pose_o, info_o = oracle.match_scan2scan(*batch[b], guesses[b])
        assert rc == status[b] and list(info[b].n_plane) == list(info_o.n_plane)
        if rc == 0:
            assert max(synth.pose_error(poses[b], pose_o)) < TIGHT


def test_empty_and_nonfinite_clouds(gpu, oracle):
    """Empty previous-scan clouds (the reference never guards them, odometry_scan_matcher.cc:57-61) give
    'too few correspondences'; NaN points must not poison neighbours, on any of the three kernels."""
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    base = list(_clouds(*_pair(oracle, 1)))
    e4, e1 = np.zeros((0, 4), np.float32), np.zeros(0, np.uint16)
    s, p, _ = gpu.match_scan2scan(e4, e1, e4, e1, base[4], base[5], ident)
    assert s == capi.TOO_FEW_CORRESPONDENCES and np.array_equal(p, ident)
    rc, po, _ = oracle.match_scan2scan(e4, e1, e4, e1, base[4], base[5], ident)
    assert rc == 1 and np.array_equal(po, ident)
    s, p, _ = gpu.match_scan2scan(base[0], base[1], base[2], base[3], e4, e4, ident)      # nothing to match
    assert s == capi.TOO_FEW_CORRESPONDENCES and np.array_equal(p, ident)
    # NaN in a target and in a query: those points never match, everything else is unaffected by the path taken
    nan_t = [np.copy(a) for a in base]
    nan_t[2][100, 1] = np.nan
    nan_q = [np.copy(a) for a in base]
    nan_q[5][7, 0] = np.nan
    variants = [base, nan_t, nan_q]
    pairs = [variants[i % 3] for i in range(45)]
    guesses = np.stack([ident] * 45)
    guesses[:, 0] = np.linspace(-0.2, 0.2, 45)
    poses, status, info = gpu.match_scan2scan_batch(_batch_sets(pairs), guesses, want_info=True)
    for b in range(45):
        s, p, i1 = gpu.match_scan2scan(*pairs[b], guesses[b])
        assert s == status[b] == 0 and np.array_equal(p, poses[b]), b
        assert np.all(np.isfinite(p))
    assert info[2].n_plane[0] in (info[0].n_plane[0], info[0].n_plane[0] - 1)         # the NaN query drops out, nothing else


@pytest.mark.parametrize("seed", range(int(os.environ.get("MSFL_FUZZ_SEEDS", "6"))))
def test_randomised_pairs_three_paths_agree(gpu, oracle, seed, monkeypatch):
    """Differential fuzzing of stage B: previous-scan clouds thinned per ring, rings removed, a block of the
    cloud moved out of order, large initial offsets.  Single-pair call (column grid, or the wave kernel for a small
    previous cloud), the same call forced onto the wave kernel, 45-pair batch (tiled edges + column-grid planes or the
    brute-force fallback) and the oracle must agree."""
    rng = np.random.default_rng(4000 + seed)
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    base = list(_clouds(*_pair(oracle, seed % 4)))
    c = [np.copy(a) for a in base]
    for k in (0, 2):                                              # thin the previous scan's clouds
        keep = rng.uniform(size=len(c[k])) < rng.uniform(0.2, 1.0)
        drop_ring = rng.integers(0, 16)
        keep &= c[k + 1] != drop_ring
        c[k], c[k + 1] = c[k][keep], c[k + 1][keep]
    if seed % 2:                                                  # a block out of ring order -> brute-force planes
        n = len(c[2]); a, b = sorted(rng.integers(0, n, 2))
        order = np.concatenate([np.arange(a, b), np.arange(0, a), np.arange(b, n)])
        c[2], c[3] = c[2][order], c[3][order]
    sel = rng.uniform(size=len(c[5])) < rng.uniform(0.3, 1.0)
    c[5] = c[5][sel]
    guess = ident.copy()
    guess[:3] = rng.normal(0, [0.5, 0.5, 0.05])
    yaw = rng.normal(0, 0.02)
    guess[5], guess[6] = np.sin(yaw / 2), np.cos(yaw / 2)
    rc, pose_o, info_o = oracle.match_scan2scan(*c, guess)
    s, pose_g, info_g = gpu.match_scan2scan(*c, guess)
    assert s == rc
    assert list(info_g.n_edge) == list(info_o.n_edge) and list(info_g.n_plane) == list(info_o.n_plane)
    assert max(synth.pose_error(pose_g, pose_o)) < TIGHT
    monkeypatch.setenv("MSFL_ODOM_WAVE_MAX_TARGETS", "1000000000")
    hw = capi.Handle(0)
    try:
        sw, pose_w, _ = hw.match_scan2scan(*c, guess)
    finally:
        hw.close()
    assert sw == s and np.array_equal(pose_w, pose_g)
    pairs = [c] * 45
    guesses = np.stack([guess] * 45)
    poses, status, _ = gpu.match_scan2scan_batch(_batch_sets(pairs), guesses)
    assert np.all(status == s) and all(np.array_equal(poses[b], pose_g) for b in (0, 17, 44))
    monkeypatch.setenv("MSFL_ODOM_BIN_SPLIT", "1")                # the index build of few long clouds: 64 workgroups per cloud
    hs = capi.Handle(0)
    try:
        poses_s, status_s, _ = hs.match_scan2scan_batch(_batch_sets(pairs), guesses)
    finally:
        hs.close()
    assert np.array_equal(status_s, status) and np.array_equal(poses_s, poses)


def test_large_gate_keeps_the_batch_path_exact(oracle):
    """the column-grid walk is exhaustive only for gates below 36 m^2.  With
    odom_distance_sq_threshold = 64 the batch call must leave the grid and still reproduce the
    single-pair (exhaustive wave kernel) results bit for bit, with guesses 6-7.5 m off so that every
    accepted correspondence lies beyond the grid's 13 x 13 column reach."""
    prm = capi.default_params()
    prm.odom_distance_sq_threshold = 64.0
    h = capi.Handle(0, prm)
    try:
        ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
        base = [_clouds(*_pair(oracle, i)) for i in range(3)]
        pairs = [base[i % 3] for i in range(45)]
        guesses = np.stack([ident] * 45)
        guesses[:, 0] = np.linspace(6.0, 7.5, 45)
        sets = _batch_sets(pairs)
        assert 45 * max(len(p[4]) + len(p[5]) for p in pairs) > 16384          # the throughput path
        poses, status, info = h.match_scan2scan_batch(sets, guesses, want_info=True)
        n_corr = 0
        for b in (0, 1, 2, 17, 31, 44):
            s, p, i1 = h.match_scan2scan(*pairs[b], guesses[b])
            assert s == status[b], b
            assert np.array_equal(p, poses[b]), b
            assert i1.n_plane[0] == info[b].n_plane[0] and i1.n_edge[0] == info[b].n_edge[0], b
            n_corr += i1.n_plane[0] + i1.n_edge[0]
        assert n_corr > 0                                                       # the wide gate really found something
    finally:
        h.close()
