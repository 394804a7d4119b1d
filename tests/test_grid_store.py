"""N1: the local map store (HybridGrid, hybrid_grid.cc:462-534).
CPU: the oracle against an independent numpy formulation.  GPU: msfl_grid_* against the oracle,
bit for bit (same cell assignment, same voxel order, same f32 accumulation order)."""
import os

import numpy as np
import pytest

from msf_loam_amd import synth
from tests import common


def _np_grid(points_batches, resolution, leaf):
    """Independent formulation: per cell, repeatedly voxel-filter (old centroids + new points)."""
    cells = {}
    for pts in points_batches:
        idx = np.stack([np.round(np.float64(pts[:, a] / np.float32(resolution))) for a in range(3)], axis=1).astype(int)
        # lround = half away from zero
        frac = (pts[:, :3] / np.float32(resolution)).astype(np.float64)
        idx = np.where(np.abs(frac - np.trunc(frac)) == 0.5, np.trunc(frac) + np.sign(frac), np.round(frac)).astype(int)
        touched = set()
        for p, c in zip(pts, map(tuple, idx)):
            cells.setdefault(c, []).append(p)
            touched.add(c)
        for c in touched:
            cells[c] = list(synth.voxel_downsample_np(np.array(cells[c], np.float32), leaf))
    order = sorted(cells, key=lambda c: (c[2], c[1], c[0]))
    return np.array([p for c in order for p in cells[c]], np.float32), order


def _batches(n_batches=4, seed=3):
    w, _, _ = common.small_world(20000)
    poses = synth.random_poses(n_batches, synth.SEED + seed)
    out = []
    for i in range(n_batches):
        pts, ring = synth.make_scan(w, poses[i], synth.SEED + 50 + i, n_az=600)
        world_pts = pts.copy()
        world_pts[:, :3] = (pts[:, :3].astype(np.float64) @ synth.quat_to_matrix(poses[i][3:]).T + poses[i][:3]).astype(np.float32)
        world_pts[:, 3] = np.linspace(0, 0.1, len(pts), dtype=np.float32)
        out.append((pts, world_pts, poses[i]))
    return out


def test_oracle_grid_matches_numpy_formulation(oracle):
    g = oracle.HybridGrid(3.0, 0.4)
    bs = _batches()
    for _, wp, _ in bs:
        assert g.insert_scan(wp) == 0
    want, order = _np_grid([wp for _, wp, _ in bs], 3.0, 0.4)
    got = g.dump()
    assert got.shape == want.shape and g.size() == (len(want), len(order))
    assert np.allclose(got, want, atol=2e-5)                 # f32 vs f64 centroid accumulation
    # re-inserting nothing changes nothing; filtering is idempotent
    g.insert_scan(np.zeros((0, 4), np.float32))
    assert np.array_equal(g.dump(), got)
    # surrounded cloud: subset of the map, whole cells, includes the cell under the sensor
    scan, _, pose = bs[0]
    s = g.get_surrounded(scan, pose)
    assert 0 < len(s) <= len(got)
    cell = lambda p: tuple(np.round(p[:3] / 3.0).astype(int))
    assert cell(np.r_[pose[:3]]) in {cell(p) for p in s[::7]} or len(s) > 1000
    # out of range -> 7, cells at half-integer boundaries round away from zero
    assert oracle.HybridGrid(3.0, 0.4).insert_scan(np.array([[3e4, 0, 0, 0]], np.float32)) == 7
    g2 = oracle.HybridGrid(3.0, 0.4)
    g2.insert_scan(np.array([[1.5, -1.5, 4.5, 0], [1.4999, -1.4999, 0, 0]], np.float32))
    assert g2.size()[1] == 2


@pytest.mark.gpu
def test_gpu_grid_matches_oracle_bit_for_bit(gpu, oracle):
    from msf_loam_amd import capi
    for leaf in (0.2, 0.4):
        go, gg = oracle.HybridGrid(3.0, leaf), capi.Grid(gpu, 3.0, leaf)
        bs = _batches()
        for k, (scan, wp, pose) in enumerate(bs):
            assert go.insert_scan(wp) == 0
            gg.insert_scan(wp)
            assert gg.size() == go.size()
            assert np.array_equal(gg.dump(), go.dump()), (leaf, k)
            for sc, ps in ((scan, pose), (bs[0][0], bs[0][2])):
                a, b = gg.get_surrounded(sc, ps), go.get_surrounded(sc, ps)
                assert np.array_equal(a, b)
        # far scan: nothing around it
        far = bs[0][0].copy(); far[:, :3] *= 0.01
        assert len(gg.get_surrounded(far, np.array([5e3, 0, 0, 0, 0, 0, 1.0]))) == 0
        # out-of-range point: MSFL_CAPACITY and the map is untouched
        before = gg.dump()
        assert gg.insert_scan(np.array([[3e4, 0, 0, 0]], np.float32), allow=(capi.CAPACITY,)) == capi.CAPACITY
        assert np.array_equal(gg.dump(), before)
        gg.close()


@pytest.mark.gpu
def test_refiltering_a_touched_cell_uses_the_centroids_coordinates(gpu, oracle):
    """HybridGrid::InsertScan re-runs the voxel filter over every cell the scan touched (hybrid_grid.cc:513-520): an old
    centroid is binned by its COORDINATES again, and an f32 centroid can round onto the next voxel's boundary (three
    points just below x = 1.4 with leaf 0.2 average to exactly 1.4f).  In a touched cell it must then merge with that
    voxel's content; in a cell the scan does not touch nothing may move; and it stays in its 3 m cell container either
    way.  Checked against the oracle (which filters each touched cell's own cloud like the reference)."""
    from msf_loam_amd import capi
    xs = np.array([1.3999997, 1.3999999, 1.3999999], np.float32)
    c = np.float32(np.float32(np.float32(xs[0] + xs[1]) + xs[2]) / np.float32(3))
    assert np.floor(xs * np.float32(5)).max() == 6 and np.floor(c * np.float32(5)) == 7      # the premise: the centroid leaves its voxel
    def blob(x0):
        p = np.zeros((4, 4), np.float32)
        p[:3, 0] = xs + np.float32(x0); p[3, 0] = np.float32(1.45) + np.float32(x0)          # three points of voxel 6, one of voxel 7
        p[:, 1] = 0.1; p[:, 2] = 0.1; p[:, 3] = [0.01, 0.02, 0.03, 0.04]
        return p
    first = np.concatenate([blob(0.0), blob(0.0) + np.array([0, 6.0, 0, 0], np.float32)])       # the same blob in cells (0,0,0) and (0,2,0)
    second = np.array([[0.5, 0.3, 0.2, 0.05]], np.float32)                                       # touches cell (0,0,0) only
    go, gg = oracle.HybridGrid(3.0, 0.2), capi.Grid(gpu, 3.0, 0.2)
    for scan in (first, second, second + np.array([0.2, 0, 0, 0], np.float32)):
        assert go.insert_scan(scan) == 0
        gg.insert_scan(scan)
        assert gg.size() == go.size()
        assert np.array_equal(gg.dump(), go.dump())
    d = go.dump()
    assert np.sum(np.abs(d[:, 1] - 0.1) < 1e-3) == 1 and np.sum(np.abs(d[:, 1] - 6.1) < 1e-3) == 2   # merged where touched, apart where not
    gg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_per_cell", [3000, 9000, 40000])
def test_dense_cells_take_every_form_of_the_rebuild(gpu, oracle, n_per_cell):
    """A 64-beam sweep puts thousands of points into one 3 m cell, a corridor wall hundreds into one 0.4 m voxel.  3 000 points per
    cell stay in the LDS form of the rebuild (points staged in LDS, run r summed by thread r), 9 000 take the large form (sort in LDS,
    wavefront sums for long runs), 40 000 its global-scratch sort; one voxel of each cloud holds 1 500 of the points (one long run).
    Inserted twice (the second insert merges with the stored centroids), plus a sparse insert that touches the same cells: every dump
    must equal the oracle's bit for bit."""
    from msf_loam_amd import capi
    rng = np.random.default_rng(77 + n_per_cell)
    def cloud(shift):
        cells = np.array([[0.0, 0.0, 0.0], [6.0, -3.0, 0.0], [-9.0, 3.0, 3.0]])
        pts = []
        for c in cells:
            p = np.zeros((n_per_cell, 4), np.float32)
            p[:, :3] = c + rng.uniform(-1.45, 1.45, (n_per_cell, 3)) * np.array([1.0, 1.0, 0.3]) + shift
            p[:1500, :3] = c + np.array([0.21, 0.21, 0.05]) + rng.uniform(0, 0.15, (1500, 3)) + shift      # one voxel, in arrival order at the front
            p[:, 3] = rng.uniform(0, 0.1, n_per_cell)
            pts.append(p)
        allp = np.concatenate(pts)
        return allp[rng.permutation(len(allp))] if n_per_cell < 40000 else allp     # shuffled: arrival order interleaves the voxels
    go, gg = oracle.HybridGrid(3.0, 0.4), capi.Grid(gpu, 3.0, 0.4)
    sparse = np.array([[0.3, 0.2, 0.1, 0.5], [6.2, -3.1, 0.0, 0.5], [-9.0, 3.0, 3.1, 0.5]], np.float32)
    for scan in (cloud(0.0), cloud(0.013), sparse):
        assert go.insert_scan(scan) == 0
        gg.insert_scan(scan)
        assert gg.size() == go.size()
        assert np.array_equal(gg.dump(), go.dump())
    gg.close()


@pytest.mark.gpu
def test_surrounded_cloud_feeds_set_map_on_device(gpu, oracle):
    """insert -> get_surrounded (device) -> msfl_set_map (device) -> match: the mapping loop without
    the map ever leaving the GPU, against the oracle doing the same through host arrays."""
    import torch
    from msf_loam_amd import capi
    w, _, _ = common.small_world(20000)
    grids = {k: (capi.Grid(gpu, 3.0, leaf), oracle.HybridGrid(3.0, leaf)) for k, leaf in (("c", 0.2), ("s", 0.4))}
    poses = synth.random_poses(6, synth.SEED + 7)
    feats = []
    for i in range(6):
        pts, ring = synth.make_scan(w, poses[i], synth.SEED + 70 + i)
        f = oracle.extract_features(pts, ring)
        feats.append(f)
        for k, key in (("c", "less_sharp"), ("s", "less_flat")):
            cloud = oracle.voxel_grid(f["full"][f[key]], 0.2 if k == "c" else 0.4)
            wp = cloud.copy()
            wp[:, :3] = (cloud[:, :3].astype(np.float64) @ synth.quat_to_matrix(poses[i][3:]).T + poses[i][:3]).astype(np.float32)
            grids[k][0].insert_scan(wp); grids[k][1].insert_scan(wp)
    f = feats[2]
    guess = synth.perturb_pose(poses[2], np.random.default_rng(1), 0.1, 1.0)
    dev = torch.device("cuda", 0)
    maps = {}
    for k, key in (("c", "less_sharp"), ("s", "less_flat")):
        scan = torch.from_numpy(np.ascontiguousarray(f["full"][f[key]])).to(dev)
        cap = grids[k][0].size()[0]
        out = torch.empty((cap, 4), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        n = grids[k][0].get_surrounded_device(scan, len(scan), guess, out, cap)
        maps[k] = (out, n, grids[k][1].get_surrounded(f["full"][f[key]], guess))
        assert n == len(maps[k][2]) and np.array_equal(out[:n].cpu().numpy(), maps[k][2])
    gpu.set_map(maps["c"][0], maps["s"][0], maps["c"][1], maps["s"][1], capi.MEM_DEVICE)
    corner, surf = oracle.voxel_grid(f["full"][f["less_sharp"]], 0.2), oracle.voxel_grid(f["full"][f["less_flat"]], 0.4)
    s, pg, _ = gpu.match_scan2map(corner, surf, guess)
    rc, po, _ = oracle.match_scan2map(maps["c"][2], maps["s"][2], corner, surf, guess)
    assert s == rc == 0 and max(synth.pose_error(pg, po)) < 1e-7
    for k in grids:
        grids[k][0].close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("MSFL_FUZZ_SEEDS", "5"))))
def test_grid_store_randomised(gpu, oracle, seed):
    """Differential fuzzing of the map store: random clouds with negative coordinates, points exactly on cell
    and voxel boundaries, duplicates and empty inserts, interleaved with surrounded-cloud queries at random
    poses; every dump and query must equal the oracle's bit for bit."""
    from msf_loam_amd import capi
    rng = np.random.default_rng(9000 + seed)
    leaf = float(rng.choice([0.2, 0.4, 1.0]))
    go, gg = oracle.HybridGrid(3.0, leaf), capi.Grid(gpu, 3.0, leaf)
    for step in range(6):
        n = int(rng.integers(0, 3000)) if step != 2 else 0
        pts = np.zeros((n, 4), np.float32)
        pts[:, :3] = rng.uniform(-40, 40, (n, 3)) * np.array([1, 1, 0.15])
        k = n // 6
        if k:
            pts[:k, :3] = np.round(pts[:k, :3] / 1.5) * 1.5              # cell boundaries (lround at .5) and voxel boundaries
            pts[k:2 * k, :3] = np.round(pts[k:2 * k, :3] / leaf) * leaf
            pts[2 * k:3 * k] = pts[:k]                                   # duplicates
        pts[:, 3] = rng.uniform(0, 0.1, n)
        assert go.insert_scan(pts) == 0
        gg.insert_scan(pts)
        assert gg.size() == go.size(), (seed, step)
        assert np.array_equal(gg.dump(), go.dump()), (seed, step)
        q = np.zeros((int(rng.integers(1, 400)), 4), np.float32)
        q[:, :3] = rng.uniform(-30, 30, (len(q), 3)) * np.array([1, 1, 0.1])
        q[0, :3] = [70, 0, 0]                                            # beyond the 60 m cut (:474)
        pose = synth.random_poses(1, 9100 + 10 * seed + step)[0]
        assert np.array_equal(gg.get_surrounded(q, pose), go.get_surrounded(q, pose)), (seed, step)
    gg.close()
