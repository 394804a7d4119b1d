"""msfl_match_pairs_batch: P (map, scan) pairs with P DIFFERENT maps in one call ("many map-submap pairs", BASELINE north star).
The reference analogue is one MappingScanMatcher::MatchScan2Map per pair, each rebuilding both kd-trees
(mapping_scan_matcher.cc:66-73; laser_mapping.cc:304-311).  Parity: equal to P single calls bit for bit, and to the oracle."""
import numpy as np
import pytest

from msf_loam_amd import synth
from tests import common

pytestmark = pytest.mark.gpu


def _pairs(oracle, P, rng, target=30000, direct=False):
    """P worlds (different pole layouts / map noise), one scan in each; every third map thinned and with a box cut out."""
    maps_c, maps_s, cs, ss, guesses, truths = [], [], [], [], [], []
    for p in range(P):
        w = synth.World(seed=synth.SEED + 700 + p, ground_half=synth.ground_half_for_target(target))
        mc, ms = synth.make_map(w, seed=synth.SEED + 1700 + p)
        if p % 3 == 1:
            ms = np.ascontiguousarray(ms[rng.uniform(size=len(ms)) < 0.6])
            lo = rng.uniform(-10, 5, 3); hi = lo + rng.uniform(3, 9, 3)
            ms = np.ascontiguousarray(ms[~np.all((ms[:, :3] > lo) & (ms[:, :3] < hi), axis=1)])
        truth = synth.random_poses(1, synth.SEED + 2700 + p)[0]
        guess = synth.perturb_pose(truth, rng)
        if direct:
            pts, ring, kind = synth.make_scan(w, truth, synth.SEED + 3700 + p, with_kind=True)
            c, s = synth.direct_features(pts, kind)
        else:
            pts, ring = synth.make_scan(w, truth, synth.SEED + 3700 + p)
            _, c, s = common.features_from_oracle(oracle, pts, ring)
        maps_c.append(mc); maps_s.append(ms); cs.append(c); ss.append(s); guesses.append(guess); truths.append(truth)
    return maps_c, maps_s, cs, ss, np.array(guesses), np.array(truths)


def _cat(lists, lead=0):
    off = np.cumsum([lead] + [len(a) for a in lists]).astype(np.int32)
    pad = np.zeros((lead, 4), np.float32)
    return np.concatenate([pad] + list(lists)), off


def test_pairs_batch_equals_single_calls_and_the_oracle(gpu, oracle):
    """8 pairs with 8 different maps: poses, statuses, accepted counts and iteration counts equal eight msfl_set_map +
    msfl_match_scan2map calls bit for bit, and the oracle within 1e-7.  Offsets that do not start at 0, a pair whose corner
    map has fewer than 5 points (MSFL_MAP_TOO_SMALL, pose untouched), a pair with an empty surf cloud."""
    from msf_loam_amd import capi
    rng = np.random.default_rng(11)
    mcs, mss, cs, ss, guesses, truths = _pairs(oracle, 8, rng)
    mcs[5] = mcs[5][:3]                                   # too few corner map points
    ss[6] = ss[6][:0]                                     # no surf features in this scan
    mc, mco = _cat(mcs, lead=17); ms, mso = _cat(mss, lead=5)
    c, co = _cat(cs, lead=3); s, so = _cat(ss)
    poses, status, info = gpu.match_pairs_batch(mc, mco, ms, mso, c, co, s, so, guesses, want_info=True)
    single = capi.Handle(0)
    for p in range(8):
        if p == 5:
            assert status[p] == capi.MAP_TOO_SMALL and np.array_equal(poses[p], guesses[p]) and list(info[p].lm_iterations) == [0, 0]
            continue
        single.set_map(mcs[p], mss[p])
        st, pose1, info1 = single.match_scan2map(cs[p], ss[p], guesses[p])
        assert st == status[p] == 0
        assert np.array_equal(pose1, poses[p]), p
        assert list(info1.n_edge) == list(info[p].n_edge) and list(info1.n_plane) == list(info[p].n_plane)
        assert list(info1.lm_iterations) == list(info[p].lm_iterations) and list(info1.final_cost) == list(info[p].final_cost)
        rc, pose_o, info_o = oracle.match_scan2map(mcs[p], mss[p], cs[p], ss[p], guesses[p])
        assert rc == 0 and list(info_o.n_edge) == list(info1.n_edge) and list(info_o.n_plane) == list(info1.n_plane)
        dt, dr = synth.pose_error(poses[p], pose_o)
        assert dt < 1e-7 and dr < 1e-7, (p, dt, dr)
        if p != 6:
            assert synth.pose_error(poses[p], truths[p])[0] < 0.05
    single.close()
    # the handle's single resident map is gone after a pairs call; set_map brings it back
    with pytest.raises(capi.MsflError) as e:
        gpu.match_scan2map(cs[0], ss[0], guesses[0])
    assert e.value.status == capi.NO_MAP
    gpu.set_map(mcs[0], mss[0])
    assert np.array_equal(gpu.match_scan2map(cs[0], ss[0], guesses[0])[1], poses[0])
    # argument errors
    with pytest.raises(capi.MsflError) as e:
        gpu.match_pairs_batch(mc, mco[::-1].copy(), ms, mso, c, co, s, so, guesses)
    assert e.value.status == capi.BAD_ARG


def test_256_pairs_with_256_different_maps(gpu, oracle):
    """The shape of the VERDICT r03 item: 256 pairs, 256 different ~30 k-point maps (7.7 M map points, one grid each), equal to
    256 single calls bit for bit, three of them against the oracle, permutation of the pairs permutes the results."""
    from msf_loam_amd import capi
    rng = np.random.default_rng(12)
    P = 256
    mcs, mss, cs, ss, guesses, truths = _pairs(oracle, P, rng, direct=True)
    mc, mco = _cat(mcs); ms, mso = _cat(mss); c, co = _cat(cs); s, so = _cat(ss)
    assert len(ms) + len(mc) > 5_000_000
    poses, status, _ = gpu.match_pairs_batch(mc, mco, ms, mso, c, co, s, so, guesses)
    assert np.all(status == 0)
    err = np.array([synth.pose_error(poses[p], truths[p]) for p in range(P)])
    assert err[:, 0].max() < 0.05 and err[:, 1].max() < 0.01
    single = capi.Handle(0)
    for p in range(P):
        single.set_map(mcs[p], mss[p])
        assert np.array_equal(single.match_scan2map(cs[p], ss[p], guesses[p])[1], poses[p]), p
    single.close()
    for p in (0, 101, 255):
        rc, pose_o, _ = oracle.match_scan2map(mcs[p], mss[p], cs[p], ss[p], guesses[p])
        dt, dr = synth.pose_error(poses[p], pose_o)
        assert rc == 0 and dt < 1e-7 and dr < 1e-7
    perm = rng.permutation(P)
    mc2, mco2 = _cat([mcs[p] for p in perm]); ms2, mso2 = _cat([mss[p] for p in perm])
    c2, co2 = _cat([cs[p] for p in perm]); s2, so2 = _cat([ss[p] for p in perm])
    poses2, status2, _ = gpu.match_pairs_batch(mc2, mco2, ms2, mso2, c2, co2, s2, so2, guesses[perm])
    assert np.array_equal(poses2, poses[perm]) and np.all(status2 == 0)
