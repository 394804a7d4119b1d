"""GPU parity of stage A (feature extraction) and the voxel helper against the CPU oracle.
Integer/index outputs must be BIT-EXACT; curvature is f32-exact; relative time within 1 ulp(f32)
(device atan2 vs libm)."""
import os

import numpy as np
import pytest

from msf_loam_amd import capi, synth
from tests import common

pytestmark = pytest.mark.gpu


def _check(f, fo):
    assert f["rc"] == fo["rc"]
    assert np.array_equal(f["ring"], fo["ring"])
    assert np.array_equal(f["full"][:, :3], fo["full"][:, :3])
    assert np.array_equal(f["curvature"], fo["curvature"])
    assert np.array_equal(f["label"], fo["label"])
    for k in ("sharp", "less_sharp", "flat", "less_flat"):
        assert np.array_equal(f[k], fo[k]), k
    dt = np.abs(f["full"][:, 3].astype(np.float64) - fo["full"][:, 3].astype(np.float64))
    assert dt.max() <= np.spacing(np.float32(0.2)), dt.max()
    assert np.mean(f["full"][:, 3] == fo["full"][:, 3]) > 0.99


def test_single_scan_bit_exact(gpu, oracle):
    for pts, ring, _, _ in common.scans(3):
        _check(gpu.extract_features(pts, ring), oracle.extract_features(pts, ring))


def test_batch_ragged_bit_exact(gpu, oracle):
    sc = common.scans(5)
    # ragged: different sizes, one shuffled-ring-order scan, one with invalid points
    clouds = []
    for i, (pts, ring, _, _) in enumerate(sc):
        p, r = pts.copy(), ring.copy()
        if i == 1:
            p, r = p[: len(p) // 2], r[: len(r) // 2]
        if i == 2:
            p[100:140, 0] = np.nan; p[500:520, :3] = 0.05
        if i == 3:
            rng = np.random.default_rng(3)
            keep = rng.uniform(size=len(p)) > 0.3          # drop points -> ragged rings
            p, r = p[keep], r[keep]
        clouds.append((p, r))
    off = np.cumsum([0] + [len(p) for p, _ in clouds]).astype(np.int32)
    res = gpu.extract_features_batch(np.concatenate([p for p, _ in clouds]), np.concatenate([r for _, r in clouds]), off)
    for (p, r), f in zip(clouds, res):
        _check(f, oracle.extract_features(p, r))


def test_minimum_range_edge_is_the_reference_compare(gpu, oracle):
    """Invalid-point removal is `norm() < min_range` on the f32 norm, compared as double (msf_loam_node.cc:85-111).  The
    kernel tests the squared norm against a threshold the host searches for; points whose norm sits within a few ulps of
    the range on either side, a squared norm that overflows (finite coordinates: the reference keeps the point) and
    non-finite coordinates must be kept / dropped exactly as the oracle does, for several ranges."""
    pts, ring, _, _ = common.scans(1)[0]
    rng = np.random.default_rng(77)
    for min_range in (0.3, 0.30000001192092896, 1.0, 2.5, 1e-3):
        p = pts.copy()
        pick = rng.choice(len(p), 4000, replace=False)
        d = p[pick, :3].astype(np.float64)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        scale = np.float32(min_range) * (1.0 + rng.integers(-6, 7, len(pick)) * 2.0 ** -24)      # a few ulps around the range
        p[pick, :3] = (d * scale[:, None]).astype(np.float32)
        p[pick[:3], 0] = [np.inf, -np.inf, np.nan]
        p[pick[3], :3] = [2e19, 1.0, 1.0]                                 # x*x overflows f32: the norm is inf, not "< min_range"
        fo = oracle.extract_features(p, ring, min_range=min_range)
        prm = capi.default_params()
        prm.min_range = min_range
        h = capi.Handle(0, params=prm)
        try:
            f = h.extract_features(p, ring)
        finally:
            h.close()
        assert f["rc"] == fo["rc"] == 0
        n_near = int(np.sum(np.abs(np.linalg.norm(p[pick[4:], :3].astype(np.float64), axis=1) / min_range - 1) < 1e-6))
        assert n_near > 3000 and 0 < len(pts) - len(fo["full"]) < len(pick)      # the edge really splits the planted points
        assert np.array_equal(f["ring"], fo["ring"]) and np.array_equal(f["full"][:, :3], fo["full"][:, :3])
        assert np.array_equal(f["label"], fo["label"])


def test_relative_time_wraps_wherever_they_fall(gpu, oracle):
    """The +2 pi wrap of the relative time (msf_loam_node.cc:145-149) is found before the scatter when it falls into the
    cloud's first 256 points and re-derived from the stored coordinates otherwise.  Full-size scans in driver order, grouped
    by ring (a ring then starts thousands of points into the cloud), started mid-revolution (the wrap lands in the middle
    of every ring) and reversed must all reproduce the oracle's times."""
    pts, ring, _, _ = common.scans(1)[0]
    n = len(pts)
    order = np.argsort(ring, kind="stable")
    variants = [(pts, ring), (pts[order], ring[order])]
    for shift in (n // 3, n // 2 + 17):
        variants.append((np.roll(pts, -shift, axis=0), np.roll(ring, -shift)))
        variants.append((np.roll(pts[order], -shift, axis=0), np.roll(ring[order], -shift)))
    variants.append((pts[::-1].copy(), ring[::-1].copy()))
    wrapped = 0
    for p, r in variants:
        fo = oracle.extract_features(p, r)
        _check(gpu.extract_features(p, r), fo)
        wrapped += int(np.sum(fo["full"][:, 3] > 0.1))
    assert wrapped > n                                                  # the wrapped times (> one scan period) really occur
    off = np.cumsum([0] + [len(p) for p, _ in variants]).astype(np.int32)
    res = gpu.extract_features_batch(np.concatenate([p for p, _ in variants]), np.concatenate([r for _, r in variants]), off)
    for (p, r), f in zip(variants, res):
        _check(f, oracle.extract_features(p, r))


def test_ring_split_over_several_workgroups_equals_the_one_workgroup_kernel(oracle, monkeypatch):
    """A call with few scans splits every scan's ring split over G workgroups (three launches; the SLAM step's one 64-beam
    sweep kept ONE compute unit busy for 326 us); a batch that fills the chip keeps the one-workgroup kernel.  Every G must
    write the same bytes -- relative times included, where the oracle comparison allows 1 ulp -- on driver-order, ring-grouped,
    mid-revolution and reversed clouds, a 64-beam sweep, tiny clouds (most slices empty), an all-invalid cloud and a bad ring."""
    pts, ring, _, _ = common.scans(1)[0]
    n = len(pts)
    order = np.argsort(ring, kind="stable")
    variants = [(pts, ring), (pts[order], ring[order]), (np.roll(pts, -n // 3, axis=0), np.roll(ring, -n // 3)),
                (np.roll(pts[order], -(n // 2 + 17), axis=0), np.roll(ring[order], -(n // 2 + 17))), (pts[::-1].copy(), ring[::-1].copy())]
    world = common.small_world()[0]
    variants.append(synth.make_scan(world, synth.random_poses(1, synth.SEED + 5)[0], synth.SEED + 6, n_beams=64, n_az=2048, elev=(-24.8, 2.0)))
    variants.append((pts[:37].copy(), ring[:37].copy()))
    variants.append((pts[:700].copy(), ring[:700].copy()))
    nanc = pts[:300].copy(); nanc[:, 0] = np.nan
    variants.append((nanc, ring[:300].copy()))
    badr = ring[:500].copy(); badr[123] = 200
    variants.append((pts[:500].copy(), badr))
    off = np.cumsum([0] + [len(p) for p, _ in variants]).astype(np.int32)
    cat_p, cat_r = np.concatenate([p for p, _ in variants]), np.concatenate([r for _, r in variants])
    results = {}
    for G in (1, 2, 4, 8, 16):
        monkeypatch.setenv("MSFL_PREP_SPLIT", str(G))
        h = capi.Handle(0)
        try:
            results[G] = h.extract_features_batch(cat_p, cat_r, off)
        finally:
            h.close()
    assert [f["rc"] for f in results[1]][-2:] == [capi.BAD_ARG, capi.BAD_RING]
    for G in (2, 4, 8, 16):
        for f, f1 in zip(results[G], results[1]):
            assert f["rc"] == f1["rc"]
            for k in ("full", "ring", "curvature", "label", "sharp", "less_sharp", "flat", "less_flat"):
                assert np.array_equal(f[k].view(np.uint8), f1[k].view(np.uint8)), (G, k)
    for (p, r), f in zip(variants[:-2], results[16][:-2]):
        _check(f, oracle.extract_features(p, r))


def test_hand_made_ties_and_gap_breaks(gpu, oracle):
    n = 300
    pts = np.zeros((n, 4), np.float32)
    pts[:, 0] = 5.0
    pts[:, 1] = np.arange(n) * 0.05
    pts[170, 0] = 6.5
    ring = np.zeros(n, np.uint16)
    _check(gpu.extract_features(pts, ring), oracle.extract_features(pts, ring))
    pts[:, 1] = np.arange(n) * 0.3
    _check(gpu.extract_features(pts, ring), oracle.extract_features(pts, ring))
    # perfectly flat wall: every curvature ties
    pts2 = np.zeros((400, 4), np.float32); pts2[:, 0] = 10.0; pts2[:, 1] = np.linspace(-4, 4, 400)
    _check(gpu.extract_features(pts2, np.zeros(400, np.uint16)), oracle.extract_features(pts2, np.zeros(400, np.uint16)))


def test_big_sector_uses_global_sort_fallback(gpu, oracle):
    """A ring with > 6*512 points: its sectors do not fit the on-chip candidate array and are picked by re-reading
    curvature and the suppression mask (there is no sort any more; the name is round 1's)."""
    rng = np.random.default_rng(11)
    n = 5000
    ang = -np.linspace(0, 2 * np.pi, n, endpoint=False)
    rad = 10 + rng.normal(0, 0.02, n) + (np.sin(ang * 40) > 0.95) * 1.5
    pts = np.zeros((n, 4), np.float32)
    pts[:, 0] = rad * np.cos(ang); pts[:, 1] = rad * np.sin(ang); pts[:, 2] = rng.normal(0, 0.01, n)
    _check(gpu.extract_features(pts, np.zeros(n, np.uint16)), oracle.extract_features(pts, np.zeros(n, np.uint16)))


def test_ring_capacity_is_an_error_not_a_wrong_answer(gpu, oracle):
    """8 128 points per ring is what the pick kernel's on-chip state holds: at the limit the result is the oracle's, one
    point more is MSFL_CAPACITY (in a batch: for that scan only)."""
    from msf_loam_amd import capi
    rng = np.random.default_rng(12)
    def ring_cloud(n):
        ang = -np.linspace(0, 2 * np.pi, n, endpoint=False)
        rad = 12 + rng.normal(0, 0.02, n) + (np.sin(ang * 25) > 0.9) * 1.0
        p = np.zeros((n, 4), np.float32)
        p[:, 0] = rad * np.cos(ang); p[:, 1] = rad * np.sin(ang); p[:, 2] = rng.normal(0, 0.01, n)
        return p, np.zeros(n, np.uint16)
    p, r = ring_cloud(8128)
    _check(gpu.extract_features(p, r), oracle.extract_features(p, r))
    p2, r2 = ring_cloud(8129)
    assert gpu.extract_features(p2, r2, allow=(capi.CAPACITY,))["rc"] == capi.CAPACITY
    off = np.array([0, len(p), len(p) + len(p2)], np.int32)
    res = gpu.extract_features_batch(np.concatenate([p, p2]), np.concatenate([r, r2]), off)
    assert res[0]["rc"] == 0 and res[1]["rc"] == capi.CAPACITY
    _check(res[0], oracle.extract_features(p, r))


def test_extrinsic_and_error_statuses(gpu, oracle):
    pts, ring, _, _ = common.scans(1)[0]
    ext = np.r_[0.5, -0.2, 0.1, synth.quat_from_euler(0.01, -0.02, 0.3)]
    f, fo = gpu.extract_features(pts, ring, extrinsic=ext), oracle.extract_features(pts, ring, extrinsic=ext)
    _check(f, fo)
    r = ring.copy(); r[5] = 128
    assert gpu.extract_features(pts, r, allow=(capi.BAD_RING,))["rc"] == capi.BAD_RING
    bad = pts.copy(); bad[:, :3] = np.nan
    assert gpu.extract_features(bad, ring, allow=(capi.BAD_ARG,))["rc"] == capi.BAD_ARG
    small = gpu.extract_features(pts[:8], ring[:8])
    assert small["rc"] == 0 and len(small["sharp"]) == 0 and len(small["less_flat"]) == 0 and len(small["full"]) == 8


def test_voxel_downsample_matches_oracle(gpu, oracle):
    pts, ring, _, _ = common.scans(1)[0]
    f = oracle.extract_features(pts, ring)
    for cloud, leaf in ((f["full"][f["less_sharp"]], 0.2), (f["full"][f["less_flat"]], 0.4), (f["full"], 0.4)):
        g = gpu.voxel_downsample(cloud, leaf)
        o = oracle.voxel_grid(cloud, leaf)
        assert g.shape == o.shape
        assert np.array_equal(g, o), np.abs(g - o).max()
    assert len(gpu.voxel_downsample(np.zeros((0, 4), np.float32), 0.2)) == 0


def test_batched_voxel_filter_matches_oracle_per_cloud(gpu, oracle):
    """msfl_voxel_downsample_batch: plain regions (ragged, one empty) and the index-list form that reads
    feature lists in the layout msfl_extract_features_batch produces; every cloud bit-for-bit equals
    the oracle's pcl::VoxelGrid restatement."""
    rng = np.random.default_rng(17)
    clouds = []
    for n in (4000, 0, 1, 2500, 7000):
        c = np.zeros((n, 4), np.float32)
        c[:, :3] = rng.uniform(-30, 30, (n, 3)) * np.array([1, 1, 0.1])
        c[:, 3] = rng.uniform(0, 0.1, n)
        clouds.append(c)
    off = np.cumsum([0] + [len(c) for c in clouds]).astype(np.int32)
    for leaf in (0.2, 0.4, 3.0):
        out, out_off = gpu.voxel_downsample_batch(np.concatenate(clouds), off, leaf)
        for b, c in enumerate(clouds):
            ref = oracle.voxel_grid(c, leaf) if len(c) else np.zeros((0, 4), np.float32)
            assert np.array_equal(out[out_off[b]:out_off[b + 1]], ref), (leaf, b)
    # index-list form: cloud b = full[off[b] + idx[off[b] + k]], k < count[b]
    w, _, _ = common.small_world()
    scans = [synth.make_scan(w, p, 700 + i) for i, p in enumerate(synth.random_poses(3, 41))]
    feats = [oracle.extract_features(p, r) for p, r in scans]
    off = np.cumsum([0] + [len(p) for p, _ in scans]).astype(np.int32)
    full = np.zeros((off[-1], 4), np.float32)
    idx = np.zeros(off[-1], np.int32)
    cnt = np.zeros(3, np.int32)
    for b, f in enumerate(feats):
        full[off[b]:off[b] + len(f["full"])] = f["full"]
        idx[off[b]:off[b] + len(f["less_flat"])] = f["less_flat"]
        cnt[b] = len(f["less_flat"])
    out, out_off = gpu.voxel_downsample_batch(full, off, 0.4, idx=idx, count=cnt)
    for b, f in enumerate(feats):
        assert np.array_equal(out[out_off[b]:out_off[b + 1]], oracle.voxel_grid(f["full"][f["less_flat"]], 0.4)), b
    assert np.array_equal(out[out_off[1]:out_off[2]], gpu.voxel_downsample(feats[1]["full"][feats[1]["less_flat"]], 0.4))


@pytest.mark.parametrize("seed", range(int(os.environ.get("MSFL_FUZZ_SEEDS", "12"))))
def test_randomised_odd_scans_bit_exact(gpu, oracle, seed):
    """Differential fuzzing on shapes a driver rarely produces: 1-5 rings out of 128, rings shorter than
    the 11-point curvature margin, interleaved / grouped ring order, duplicated points, NaN / inf / too-near
    points, quantised coordinates (curvature ties), a full turn that wraps the relative time."""
    rng = np.random.default_rng(1000 + seed)
    n_rings = int(rng.integers(1, 6))
    ring_ids = np.sort(rng.choice(128, n_rings, replace=False))
    n = int(rng.integers(40, 4000))
    ring = ring_ids[rng.integers(0, n_rings, n)].astype(np.uint16)
    if seed % 3 == 0:
        ring = np.sort(ring)                                          # grouped by ring instead of interleaved
    if seed % 4 == 1 and n_rings > 1:
        ring[ring == ring_ids[0]] = ring_ids[1]
        ring[: int(rng.integers(1, 10))] = ring_ids[0]                # a ring with fewer than 11 points
    az = np.sort(rng.uniform(-np.pi, np.pi, n))[::-1] if seed % 2 == 0 else np.cumsum(rng.uniform(0, 4 * np.pi / n, n)) % (2 * np.pi) - np.pi
    rad = rng.uniform(0.2, 40.0, n)
    if seed % 5 == 2:
        rad = np.round(rad, 0)                                         # lattice radii: many equal curvatures
    pts = np.zeros((n, 4), np.float32)
    pts[:, 0], pts[:, 1] = rad * np.cos(az), rad * np.sin(az)
    pts[:, 2] = rng.normal(0, 0.5, n) if seed % 2 else np.round(rng.normal(0, 0.5, n), 1)
    k = max(1, n // 50)
    idx = rng.choice(n, 4 * k, replace=False)
    pts[idx[:k], 0] = np.nan
    pts[idx[k:2 * k], 1] = np.inf
    pts[idx[2 * k:3 * k], :3] *= 0.001                                # inside min_range
    dup = idx[3 * k:]
    pts[dup] = pts[(dup + 1) % n]                                     # exact duplicates of a neighbour
    fo = oracle.extract_features(pts, ring)
    f = gpu.extract_features(pts, ring, allow=(capi.BAD_ARG,))
    _check(f, fo)


def test_voxel_filter_drops_non_finite_points_like_pcl(gpu, oracle):
    """pcl::VoxelGrid::applyFilter skips non-finite points of a non-dense cloud (`if (!input_->is_dense) if (!pcl_isfinite ...)
    continue;`): msfl_voxel_downsample does the same (device-side compaction, then the filter), so the result equals the filter
    of the finite points alone, bit for bit, whatever NaN / Inf coordinates are sprinkled in.  A cloud of nothing but such
    points gives an empty result.  The batch forms (fed by the extraction, which cannot emit one) refuse the cloud."""
    from msf_loam_amd import capi
    rng = np.random.default_rng(1)
    pts = np.zeros((5000, 4), np.float32)
    pts[:, :3] = rng.uniform(-5, 5, (5000, 3)); pts[:, 3] = rng.uniform(0, 0.1, 5000)
    clean = gpu.voxel_downsample(pts, 0.4)
    assert len(clean) > 0 and np.array_equal(clean, oracle.voxel_grid(pts, 0.4))
    for bad in (np.nan, np.inf, -np.inf):
        q = pts.copy()
        hit = rng.choice(len(q), 37, replace=False)
        q[hit, rng.integers(0, 3, 37)] = bad
        keep = np.ones(len(q), bool); keep[hit] = False
        out = gpu.voxel_downsample(q, 0.4)
        assert np.array_equal(out, oracle.voxel_grid(q[keep], 0.4))          # = PCL on the finite points, arrival order kept
        with pytest.raises(capi.MsflError) as e:
            gpu.voxel_downsample_batch(np.concatenate([pts, q]), np.array([0, 5000, 10000], np.int32), 0.4)
        assert e.value.status == capi.BAD_ARG
    allbad = pts.copy(); allbad[:, 0] = np.nan
    assert len(gpu.voxel_downsample(allbad, 0.4)) == 0
    assert np.array_equal(gpu.voxel_downsample(pts, 0.4), clean)                # and the handle is fine afterwards
    out, off = gpu.voxel_downsample_batch(np.concatenate([pts, pts]), np.array([0, 5000, 10000], np.int32), 0.4)
    assert off[1] == off[2] - off[1] == len(clean)


def test_batched_voxel_filter_lds_form_and_its_fallbacks(oracle, monkeypatch):
    """The batched filter sorts each cloud's RUNS in LDS (one workgroup per cloud) and falls back to the device-wide radix
    sort for the whole batch when a cloud does not fit (more than 65 535 points, too many runs, coordinates beyond +-8191
    voxels).  Both forms must agree with the oracle bit for bit: real feature lists (long runs), uniformly random points
    (every run has length 1), a cloud above each limit, and duplicated points."""
    from msf_loam_amd import capi
    rng = np.random.default_rng(23)
    w, _, _ = common.small_world()
    scans = [synth.make_scan(w, p, 900 + i) for i, p in enumerate(synth.random_poses(4, 43))]
    lists = []
    for p, r in scans:
        f = oracle.extract_features(p, r)
        lists.append(f["full"][f["less_flat"]])
        lists.append(f["full"][f["less_sharp"]])
    rand = np.zeros((9000, 4), np.float32); rand[:, :3] = rng.uniform(-20, 20, (9000, 3)); rand[:, 3] = rng.uniform(0, 0.1, 9000)
    dup = np.repeat(rand[:300], 7, axis=0)                                  # runs of 7 identical points
    line = np.zeros((3000, 4), np.float32); line[:, 0] = np.linspace(-50, 50, 3000)   # one long polyline: runs of ~12
    base = lists + [rand, dup, line, np.zeros((0, 4), np.float32), rand[:1]]

    def check(h, clouds, leaf):
        off = np.cumsum([0] + [len(c) for c in clouds]).astype(np.int32)
        out, out_off = h.voxel_downsample_batch(np.concatenate(clouds), off, leaf)
        for b, c in enumerate(clouds):
            ref = oracle.voxel_grid(c, leaf) if len(c) else np.zeros((0, 4), np.float32)
            assert np.array_equal(out[out_off[b]:out_off[b + 1]], ref), (leaf, b, len(c))
        return out, out_off

    h = capi.Handle(0)
    monkeypatch.setenv("MSFL_VOXEL_GLOBAL", "1")
    hg = capi.Handle(0)
    try:
        for leaf in (0.2, 0.4, 1.5):
            a = check(h, base, leaf)
            g = check(hg, base, leaf)
            assert np.array_equal(a[0], g[0]) and np.array_equal(a[1], g[1])
        big = np.zeros((70000, 4), np.float32); big[:, :3] = rng.uniform(-30, 30, (70000, 3))     # > 65 535 points
        many_runs = np.zeros((30000, 4), np.float32); many_runs[:, :3] = rng.uniform(-30, 30, (30000, 3))   # 30 000 runs of length 1
        far = rand.copy(); far[0, 0] = 5000.0                                                      # 25 000 voxels from the rest at 0.2 m
        for extra, leaf in ((big, 0.4), (many_runs, 0.4), (far, 0.2)):
            check(h, base[:3] + [extra] + base[3:6], leaf)
    finally:
        h.close(); hg.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("MSFL_FUZZ_SEEDS", "8"))))
def test_randomised_voxel_clouds_both_forms(oracle, monkeypatch, seed):
    """Differential fuzzing of the batched voxel filter: clouds made of sweeps (runs of consecutive points per voxel of
    every length, voxels that a later sweep revisits, voxels with hundreds of points), jittered duplicates, negative and
    offset coordinates, index lists with per-cloud counts, empty clouds in between.  The LDS form and the device-wide
    run sort must both equal the oracle bit for bit."""
    from msf_loam_amd import capi
    rng = np.random.default_rng(7000 + seed)
    leaf = float(rng.choice([0.1, 0.2, 0.4, 0.75]))
    clouds = []
    for _ in range(int(rng.integers(3, 9))):
        kind = int(rng.integers(0, 5))
        if kind == 0:
            clouds.append(np.zeros((0, 4), np.float32)); continue
        n_sweeps = int(rng.integers(1, 12))
        parts = []
        centre = rng.uniform(-40, 40, 3)
        for s_ in range(n_sweeps):
            m = int(rng.integers(1, 2500))
            step = rng.uniform(0.002, min(0.6, 40.0 / m))                   # long runs ... one point per voxel; extent kept below pcl's index range
            t = np.arange(m) * step
            d = rng.normal(size=3); d /= np.linalg.norm(d)
            start = centre + rng.normal(0, 1.0 if kind < 3 else 0.05, 3)     # kind 3/4: the sweeps overlap heavily (revisited voxels)
            parts.append(start[None, :] + t[:, None] * d[None, :] + rng.normal(0, 0.003, (m, 3)))
        xyz = np.concatenate(parts)
        if kind == 4:
            xyz = np.concatenate([xyz, np.repeat(xyz[:50], 6, axis=0)])     # exact duplicates, far from their originals in order
        c = np.zeros((len(xyz), 4), np.float32); c[:, :3] = xyz; c[:, 3] = rng.uniform(0, 0.1, len(xyz))
        clouds.append(c)
    # through an index list with per-cloud counts (what the pipeline does), slack between the clouds
    slack = [int(rng.integers(0, 30)) for _ in clouds]
    off = np.cumsum([0] + [len(c) + sl for c, sl in zip(clouds, slack)]).astype(np.int32)
    full = np.zeros((off[-1], 4), np.float32); idx = np.zeros(off[-1], np.int32); cnt = np.zeros(len(clouds), np.int32)
    for b, c in enumerate(clouds):
        perm = rng.permutation(len(c) + slack[b])[: len(c)] if len(c) else np.zeros(0, np.int64)
        full[off[b] + perm] = c                                              # stored scattered, listed in cloud order
        idx[off[b]: off[b] + len(c)] = perm
        cnt[b] = len(c)
    h = capi.Handle(0)
    monkeypatch.setenv("MSFL_VOXEL_GLOBAL", "1")
    hg = capi.Handle(0)
    try:
        for hh in (h, hg):
            out, out_off = hh.voxel_downsample_batch(full, off, leaf, idx=idx, count=cnt)
            for b, c in enumerate(clouds):
                ref = oracle.voxel_grid(c, leaf) if len(c) else np.zeros((0, 4), np.float32)
                assert np.array_equal(out[out_off[b]:out_off[b + 1]], ref), (seed, leaf, b, len(c))
    finally:
        h.close(); hg.close()


def test_voxel_pair_call_equals_the_two_batch_calls(gpu, oracle):
    """msfl_voxel_downsample_batch_pair on device memory (both filters enqueued before one synchronisation) against the two
    plain batch calls: feature lists of real scans with per-cloud counts, an empty list in the middle; then a batch with a
    cloud above the LDS form's point limit in list b only (that list alone is redone by the device-wide form)."""
    import ctypes as C
    import torch
    from msf_loam_amd import capi
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(31)
    sc = common.scans(5)
    feats = [oracle.extract_features(p, r) for p, r, _, _ in sc]
    def run(feats, extra_b=None, extra_a=None):
        sizes = [len(f["full"]) for f in feats]
        if extra_b is not None: sizes[2] = max(sizes[2], len(extra_b))
        if extra_a is not None: sizes[3] = max(sizes[3], len(extra_a))
        off = np.cumsum([0] + sizes).astype(np.int32)
        n = int(off[-1])
        full = np.zeros((n, 4), np.float32); ia = np.zeros(n, np.int32); ib = np.zeros(n, np.int32)
        ca = np.zeros(len(feats), np.int32); cb = np.zeros(len(feats), np.int32)
        for b, f in enumerate(feats):
            full[off[b]:off[b] + len(f["full"])] = f["full"]
            la, lb = f["less_sharp"], f["less_flat"]
            if b == 1: la = la[:0]
            if b == 2 and extra_b is not None:
                full[off[b]:off[b] + len(extra_b)] = extra_b; lb = np.arange(len(extra_b), dtype=np.int32)
            if b == 3 and extra_a is not None:
                full[off[b]:off[b] + len(extra_a)] = extra_a; la = np.arange(len(extra_a), dtype=np.int32); lb = lb[lb < 0]
            ia[off[b]:off[b] + len(la)] = la; ca[b] = len(la)
            ib[off[b]:off[b] + len(lb)] = lb; cb[b] = len(lb)
        t = lambda a: torch.from_numpy(a).to(dev)
        d_full, d_ia, d_ib, d_ca, d_cb = t(full), t(ia), t(ib), t(ca), t(cb)
        outs = [torch.zeros((n, 4), dtype=torch.float32, device=dev) for _ in range(4)]
        oo = [np.zeros(len(feats) + 1, np.int32) for _ in range(4)]
        vp = C.c_void_p
        B = len(feats)
        s = gpu.lib.msfl_voxel_downsample_batch_pair(gpu.h, C.c_int(B), vp(d_full.data_ptr()), off.ctypes.data_as(vp),
                                                     vp(d_ia.data_ptr()), vp(d_ca.data_ptr()), C.c_float(0.2), vp(outs[0].data_ptr()), oo[0].ctypes.data_as(vp),
                                                     vp(d_ib.data_ptr()), vp(d_cb.data_ptr()), C.c_float(0.4), vp(outs[1].data_ptr()), oo[1].ctypes.data_as(vp),
                                                     C.c_int(capi.MEM_DEVICE))
        assert s == 0, s
        for k, (di, dc, leaf) in enumerate(((d_ia, d_ca, 0.2), (d_ib, d_cb, 0.4))):
            s = gpu.lib.msfl_voxel_downsample_batch(gpu.h, C.c_int(B), vp(d_full.data_ptr()), vp(di.data_ptr()), off.ctypes.data_as(vp),
                                                    vp(dc.data_ptr()), C.c_float(leaf), vp(outs[2 + k].data_ptr()), oo[2 + k].ctypes.data_as(vp),
                                                    C.c_int(capi.MEM_DEVICE))
            assert s == 0, s
            assert np.array_equal(oo[k], oo[2 + k])
            m = int(oo[k][-1])
            assert m > 0 and np.array_equal(outs[k][:m].cpu().numpy(), outs[2 + k][:m].cpu().numpy())
        return oo
    oo = run(feats)
    assert oo[0][2] == oo[0][1]                                            # the empty list
    big = np.zeros((70000, 4), np.float32); big[:, :3] = rng.uniform(-25, 25, (70000, 3))
    run(feats, extra_b=big)
    # list a alone above the limit (its fallback used to overwrite list b's pending read-back), then both lists
    run(feats, extra_a=big)
    run(feats, extra_b=big, extra_a=big[:66000])


def _sweep_cloud(rng, n, extent=25.0, step=0.008, jitter=0.004):
    """n points along a random polyline of sweeps (runs of consecutive same-voxel points, voxels revisited by later sweeps)."""
    parts, left = [], n
    while left > 0:
        m = min(left, int(rng.integers(400, 6000)))
        d = rng.normal(size=3); d[2] *= 0.2; d /= np.linalg.norm(d)
        start = rng.uniform(-extent, extent, 3) * [1, 1, 0.1]
        parts.append(start[None, :] + (np.arange(m) * step)[:, None] * d[None, :] + rng.normal(0, jitter, (m, 3)))
        left -= m
    c = np.zeros((n, 4), np.float32)
    c[:, :3] = np.concatenate(parts)
    c[:, 3] = rng.uniform(0, 0.1, n)
    return c


def test_voxel_lists_beyond_the_lds_forms_and_pooled_run_slots(oracle, monkeypatch):
    """Round 5.  (i) The run slots of the LDS form are one pool per cloud: a list whose runs crowd into one wavefront's slice (the canopy
    rings of an outdoor scan) fits as long as the cloud's total does.  (ii) Lists of 65 536 .. 131 071 points (a 64-beam less-flat list)
    take the big one-workgroup form (records in global scratch, 13-bit coordinates), with the device-wide form behind it for what that
    cannot hold either: more than 131 071 points, more than 65 536 runs, coordinates beyond +-4 095 voxels.  Every path equals the oracle
    bit for bit, and the three forms equal each other (MSFL_VOXEL_NO_BIG / MSFL_VOXEL_GLOBAL handles)."""
    from msf_loam_amd import capi
    rng = np.random.default_rng(91)
    # (i) 20 000 points: 18 000 along one slow sweep (few runs), then 2 000 scattered ones (2 000 runs in the last slice of ~1 300 points)
    crowd = np.concatenate([_sweep_cloud(rng, 18000, step=0.004), np.zeros((2000, 4), np.float32)])
    crowd[18000:, :3] = rng.uniform(-25, 25, (2000, 3))
    big_a = _sweep_cloud(rng, 100_000)                         # ~100 k points, long runs: the big form
    big_b = _sweep_cloud(rng, 131_071)                         # its largest size
    over = _sweep_cloud(rng, 131_072)                          # one more: device-wide
    many = np.zeros((90_000, 4), np.float32); many[:, :3] = rng.uniform(-40, 40, (90_000, 3))    # 90 000 runs of length 1 (> 65 536 slots)
    wide = _sweep_cloud(rng, 70_000, extent=4.0, step=0.002); wide[:100, 0] += 1500.0   # 7 500 voxels from the rest at 0.2 m: beyond the big form's +-4 095
    small = _sweep_cloud(rng, 3000)
    empty = np.zeros((0, 4), np.float32)
    cases = [([crowd], 0.4), ([crowd, small], 0.2), ([big_a], 0.4), ([small, big_a, empty, big_b], 0.4), ([big_b], 0.2), ([over, small], 0.4),
             ([many, small], 0.4), ([wide, small], 0.2), ([big_a, many, crowd], 0.4)]
    hs = {}
    hs["default"] = capi.Handle(0)
    monkeypatch.setenv("MSFL_VOXEL_NO_BIG", "1"); hs["no_big"] = capi.Handle(0); monkeypatch.delenv("MSFL_VOXEL_NO_BIG")
    monkeypatch.setenv("MSFL_VOXEL_GLOBAL", "1"); hs["global"] = capi.Handle(0); monkeypatch.delenv("MSFL_VOXEL_GLOBAL")
    try:
        for clouds, leaf in cases:
            off = np.cumsum([0] + [len(c) for c in clouds]).astype(np.int32)
            outs = {k: h.voxel_downsample_batch(np.concatenate(clouds), off, leaf) for k, h in hs.items()}
            for b, c in enumerate(clouds):
                ref = oracle.voxel_grid(c, leaf) if len(c) else np.zeros((0, 4), np.float32)
                for k, (out, out_off) in outs.items():
                    assert np.array_equal(out[out_off[b]:out_off[b + 1]], ref), (k, leaf, b, len(c))
        # through the index-list form the pipeline uses (device-resident counts), big list next to a small one
        clouds = [big_a, small]
        off = np.cumsum([0] + [len(c) + 17 for c in clouds]).astype(np.int32)
        full = np.zeros((off[-1], 4), np.float32); idx = np.zeros(off[-1], np.int32); cnt = np.array([len(c) for c in clouds], np.int32)
        for b, c in enumerate(clouds):
            perm = rng.permutation(len(c) + 17)[:len(c)]
            full[off[b] + perm] = c; idx[off[b]:off[b] + len(c)] = perm
        out, out_off = hs["default"].voxel_downsample_batch(full, off, 0.4, idx=idx, count=cnt)
        for b, c in enumerate(clouds):
            assert np.array_equal(out[out_off[b]:out_off[b + 1]], oracle.voxel_grid(c, 0.4)), b
    finally:
        for h in hs.values():
            h.close()


def test_non_finite_fourth_field_stays_in_its_own_voxel_in_every_form(oracle, monkeypatch):
    """ADVICE r05.  A point with finite coordinates and a NaN / Inf 4th field (intensity / relative time) is a point pcl::VoxelGrid KEEPS:
    its voxel's 4th sum goes non-finite and no other voxel's does.  The one-workgroup forms sum a chunk's runs in lockstep with a
    1.0 / 0.0 multiplier per lane (0 * NaN = NaN would leak into the other runs of the chunk), so they hand such a cloud to the
    device-wide form, which adds point by point.  Every entry point and every form equals the oracle (NaN compared as NaN), the
    other clouds of the batch are untouched, and the finite columns are bit-identical."""
    from msf_loam_amd import capi
    rng = np.random.default_rng(77)
    small = _sweep_cloud(rng, 6000)
    big = _sweep_cloud(rng, 100_000)
    hs = {}
    hs["default"] = capi.Handle(0)
    monkeypatch.setenv("MSFL_VOXEL_NO_BIG", "1"); hs["no_big"] = capi.Handle(0); monkeypatch.delenv("MSFL_VOXEL_NO_BIG")
    monkeypatch.setenv("MSFL_VOXEL_GLOBAL", "1"); hs["global"] = capi.Handle(0); monkeypatch.delenv("MSFL_VOXEL_GLOBAL")
    try:
        for bad in (np.nan, np.inf, -np.inf):
            for base in (small, big):
                q = base.copy()
                hit = rng.choice(len(q), 23, replace=False)
                q[hit, 3] = bad
                if bad == np.inf: q[hit[0], 3] = -np.inf                      # +Inf and -Inf meeting in one sum is fine too
                ref = oracle.voxel_grid(q, 0.4)
                n_bad = int((~np.isfinite(ref[:, 3])).sum())
                assert 1 <= n_bad <= 23 and np.isfinite(ref[:, :3]).all()
                for name, h in hs.items():
                    out = h.voxel_downsample(q, 0.4)
                    assert np.array_equal(out, ref, equal_nan=True), (name, bad, len(q))
                    assert np.array_equal(out[:, :3], ref[:, :3])
                    clouds = [small, q, small[:1500]]
                    off = np.cumsum([0] + [len(c) for c in clouds]).astype(np.int32)
                    o, oo = h.voxel_downsample_batch(np.concatenate(clouds), off, 0.4)
                    for b, c in enumerate(clouds):
                        assert np.array_equal(o[oo[b]:oo[b + 1]], oracle.voxel_grid(c, 0.4), equal_nan=True), (name, bad, b)
    finally:
        for h in hs.values():
            h.close()
