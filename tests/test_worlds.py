"""The harness's other two worlds (msf_loam_amd/worlds.py, round 5): what they promise, checked on the CPU.

outdoor: leaf-dense volumes in >= 15 % of the occupied 1 m cells of the surf map, sensor returns from inside the volumes;
corridor: the along-axis direction nearly unobservable.  Both: deterministic, and the oracle registers scans in them (so the
GPU parity tests on these worlds compare two WORKING registrations, not two failures)."""
import numpy as np
import pytest

from msf_loam_amd import synth
from tests import common


def test_room_world_is_untouched_by_the_new_kinds():
    a, b = synth.World(ground_half=20.0), synth.World(ground_half=20.0, kind="room")
    assert a.geom is None and np.array_equal(a.poles, b.poles)
    p = synth.random_poses(1, 5)[0]
    pa, ra = synth.make_scan(a, p, 77)
    pb, rb = synth.make_scan(b, p, 77)
    assert np.array_equal(pa, pb) and np.array_equal(ra, rb)
    assert np.array_equal(synth.world_poses(a, 3, 9), synth.random_poses(3, 9))
    with pytest.raises(ValueError):
        synth.World(kind="cave")


@pytest.mark.parametrize("kind", ["outdoor", "corridor"])
def test_worlds_are_deterministic(kind):
    w1, w2 = synth.World(kind=kind), synth.World(kind=kind)
    m1, m2 = synth.make_map(w1), synth.make_map(w2)
    assert np.array_equal(m1[0], m2[0]) and np.array_equal(m1[1], m2[1])
    p = synth.world_poses(w1, 2, 31)
    assert np.array_equal(p, synth.world_poses(w2, 2, 31))
    a = synth.make_scan(w1, p[1], 5, with_kind=True)
    b = synth.make_scan(w2, p[1], 5, with_kind=True)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert len(a[0]) > 20000 and np.isfinite(a[0]).all()


def test_outdoor_map_is_leaf_dense_where_the_volumes_are():
    w, mc, ms = common.other_world("outdoor")
    assert 350_000 < len(ms) < 700_000 and len(mc) > 100_000
    c = np.floor(ms[:, :3]).astype(np.int64)
    key = (c[:, 0] + 500) * 1_000_000 + (c[:, 1] + 500) * 1_000 + (c[:, 2] + 100)
    _, cnt = np.unique(key, return_counts=True)
    assert (cnt >= 8).mean() >= 0.15, (cnt >= 8).mean()          # >= 8 points / m^3 (half of the 0.4 m leaf's 15.6) in >= 15 % of occupied cells
    # the surfaces stay at the leaf spacing: a ground cell holds ~6 points
    assert 5.0 < np.median(cnt) < 8.0
    # and the sensor sees INTO the volumes: a tenth or more of a scan's returns come from inside one
    pts, ring, kind = synth.make_scan(w, synth.world_poses(w, 1, 3)[0], 4, with_kind=True)
    assert 0.08 < (kind == 3).mean() < 0.5 and (kind == 0).mean() > 0.1 and (kind == 1).any() and (kind == 2).any()
    assert len(np.unique(ring)) == 16


def test_outdoor_ground_hits_lie_on_the_relief():
    w, _, _ = common.other_world("outdoor")
    p = synth.world_poses(w, 1, 17)[0]
    pts, ring, kind = synth.make_scan(w, p, 4, noise=0.0, with_kind=True)
    P = pts[kind == 0, :3].astype(np.float64) @ synth.quat_to_matrix(p[3:]).T + p[:3]
    assert np.abs(P[:, 2] - w.geom.ground(P[:, 0], P[:, 1])).max() < 1e-4
    assert np.ptp(P[:, 2]) > 0.2                               # a relief, not a plane


@pytest.mark.parametrize("kind", ["outdoor", "corridor"])
def test_the_oracle_registers_scans_in_the_other_worlds(oracle, kind):
    _, mc, ms = common.other_world(kind)
    n_rejected = 0
    for pts, ring, truth, guess in common.other_scans(kind, 3):
        _, corner, surf = common.features_from_oracle(oracle, pts, ring)
        rc, pose, info = oracle.match_scan2map(mc, ms, corner, surf, guess)
        assert rc == 0 and info.n_plane[1] > 500 and info.n_edge[1] > 100
        et, er = synth.pose_error(pose, truth)
        gt, gr = synth.pose_error(guess, truth)
        assert et < max(0.5 * gt, 0.06) and er < max(0.5 * gr, 0.02), (et, er, gt, gr)
        n_rejected += sum(info.lm_iterations) - sum(info.lm_successful)
    if kind == "corridor":
        assert n_rejected > 0                                  # trust-region steps get rejected here (they never are in the room)


def test_corridor_axis_is_nearly_unobservable(oracle):
    """The planes' normal equations along the corridor axis: of ~1 300 accepted plane correspondences only the few dozen on the
    two end walls (two rings at 10-70 m) have a normal along x, so J^T J's smallest eigenvalue is the x direction's and is
    3-15 % of the largest."""
    _, mc, ms = common.other_world("corridor")
    pts, ring, truth, guess = common.other_scans("corridor", 1)[0]
    _, corner, surf = common.features_from_oracle(oracle, pts, ring)
    corr = oracle.associate_scan2map(mc, ms, corner, surf, truth)
    pl = corr[(corr["kind"] == 2)]
    H = pl["N"].T @ pl["N"]                                   # translation block of J^T J for the plane factors (J_t = N^T)
    ev = np.linalg.eigvalsh(H)
    assert ev[0] < 0.5 * ev[1] and ev[0] < 0.15 * ev[2], ev
    assert (np.abs(pl["N"][:, 0]) > 0.7).sum() < 0.1 * len(pl)
    assert abs(np.linalg.eigh(H)[1][0, 0]) > 0.99             # and that direction is the x axis
