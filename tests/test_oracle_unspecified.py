"""VERDICT r01 'Next round' 5b: how much does the final scan-to-map pose depend on the three behaviours the reference's
toolchain leaves unspecified (and the oracle / the GPU path fix by fiat)?

    sort ties   std::sort of a sector's indices by curvature is unstable        msf_loam_node.cc:263-267
    kNN ties    FLANN's order among equal distances                            mapping_scan_matcher.cc:125
    atan2       float or double overload of unqualified atan2(float, float)    msf_loam_node.cc:131,139

The whole chain (extraction -> 0.2 / 0.4 m voxel grid -> registration with two outer iterations) is run with each
choice flipped, on the 50k-map fixtures AND on a tie-rich variant of them (scan and map coordinates snapped to a
2 cm / 5 cm lattice so that equal curvatures actually occur; equal kNN distances between DISTINCT map
points need a lattice map and on-lattice queries: second test).  Bar: the north-star tolerance,
1e-4 m / 1e-4 rad, with an order of magnitude in hand."""
import ctypes as C

import numpy as np
import pytest

from msf_loam_amd import synth
from tests import common

VARIANTS = {"sort_ties_reversed": (1, 0, 0), "knn_ties_reversed": (0, 1, 0), "atan2f": (0, 0, 1), "all_three": (1, 1, 1)}


def _pipeline(orc, mc, ms, pts, ring, guess):
    f = orc.extract_features(pts, ring)
    corner = orc.voxel_grid(f["full"][f["less_sharp"]], 0.2)
    surf = orc.voxel_grid(f["full"][f["less_flat"]], 0.4)
    rc, pose, info = orc.match_scan2map(mc, ms, corner, surf, guess)
    assert rc == 0
    return f, pose


def _snap(a, q):
    out = np.array(a, dtype=np.float32, copy=True)
    out[:, :3] = (np.round(out[:, :3].astype(np.float64) / q) * q).astype(np.float32)
    return out


@pytest.fixture
def unspecified(oracle):
    lib = oracle.lib()
    yield lambda s, k, a: lib.orc_set_unspecified(C.c_int(s), C.c_int(k), C.c_int(a))
    lib.orc_set_unspecified(C.c_int(0), C.c_int(0), C.c_int(0))


@pytest.mark.parametrize("tie_rich", [False, True], ids=["fixtures", "lattice-snapped fixtures"])
def test_pose_is_insensitive_to_the_unspecified_choices(oracle, unspecified, tie_rich):
    _, mc, ms = common.small_world()
    if tie_rich:
        mc, ms = _snap(mc, 0.05), _snap(ms, 0.05)
    changed = {k: 0 for k in VARIANTS}
    worst = {k: (0.0, 0.0) for k in VARIANTS}
    for pts, ring, truth, guess in common.scans(3):
        if tie_rich:
            pts = _snap(pts, 0.02)
        unspecified(0, 0, 0)
        f0, pose0 = _pipeline(oracle, mc, ms, pts, ring, guess)
        for name, flags in VARIANTS.items():
            unspecified(*flags)
            f1, pose1 = _pipeline(oracle, mc, ms, pts, ring, guess)
            dt, dr = synth.pose_error(pose1, pose0)
            worst[name] = (max(worst[name][0], dt), max(worst[name][1], dr))
            differs = any(not np.array_equal(f0[k], f1[k]) for k in ("sharp", "less_sharp", "flat", "less_flat")) or \
                not np.array_equal(f0["full"], f1["full"]) or not np.array_equal(pose0, pose1)
            changed[name] += int(differs)
    for name, (dt, dr) in worst.items():
        assert dt < 1e-5 and dr < 1e-5, (name, dt, dr)          # north-star bar is 1e-4: an order of magnitude in hand
    if tie_rich:
        # the flipped choices must actually have been exercised: something in the chain changed
        assert changed["atan2f"] > 0 and changed["all_three"] > 0, changed
        assert changed["sort_ties_reversed"] > 0, changed
    print({k: ("%.2e m" % v[0], "%.2e rad" % v[1], "runs that differ: %d" % changed[k]) for k, v in worst.items()})


def test_knn_tie_order_does_not_move_the_residuals(oracle, unspecified):
    """Exactly tied kNN distances between distinct map points: lattice map, queries on / between lattice nodes.  Flipping
    the tie order changes which of the tied points enter the five-neighbour sets, but tied points of a lattice plane /
    a lattice line span the same plane / line, so the fitted residual of the query does not move."""
    g = np.arange(-8, 9, dtype=np.float32) * 0.5
    X, Y = np.meshgrid(g, g, indexing="ij")
    plane = np.stack([X.ravel(), Y.ravel(), np.full(X.size, -1.5, np.float32), np.zeros(X.size, np.float32)], 1)
    line = np.stack([np.zeros(80, np.float32), np.zeros(80, np.float32), np.arange(80, dtype=np.float32) * 0.125 - 1.5, np.zeros(80, np.float32)], 1)
    rng = np.random.default_rng(3)
    ms, mc = plane[rng.permutation(len(plane))], line[rng.permutation(len(line))]
    nodes = plane[rng.integers(0, len(plane), 60)].copy(); nodes[:, 2] += 0.25          # above lattice nodes: 4-way ties at the 2nd..5th place
    mids = nodes.copy(); mids[:, 0] += 0.25                                              # between two nodes: 2-way ties everywhere
    surf = np.concatenate([nodes, mids]).astype(np.float32)
    corner = np.stack([np.full(20, 0.3, np.float32), np.zeros(20, np.float32), np.arange(20, dtype=np.float32) * 0.25 + 0.0625, np.zeros(20, np.float32)], 1)
    pose = np.array([0, 0, 0, 0, 0, 0, 1.0])
    unspecified(0, 0, 0)
    a = oracle.associate_scan2map(mc, ms, corner, surf, pose)
    unspecified(0, 1, 0)
    b = oracle.associate_scan2map(mc, ms, corner, surf, pose)
    assert np.array_equal(a["kind"], b["kind"]) and (a["kind"] != 0).sum() > 100
    ok = a["kind"] != 0
    assert np.abs(a["C"][ok] - b["C"][ok]).max() > 1e-3, "the tie order must have changed some five-neighbour sets"

    def residual(c):
        w = c["p"]                                          # identity pose
        d = w - c["C"]
        r = np.where((c["kind"] == 2)[:, None], np.sum(c["N"] * d, 1, keepdims=True) * np.ones((1, 3)) / np.sqrt(3.0), np.cross(c["N"], d))
        return np.linalg.norm(r, axis=1)
    assert np.abs(residual(a[ok]) - residual(b[ok])).max() < 1e-9
