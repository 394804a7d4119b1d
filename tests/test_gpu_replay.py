"""Sequential replay (extract -> scan2scan -> scan2map with an accumulating map) through the GPU
library vs the same loop driven by the CPU oracle: pose-by-pose parity and a bounded drift."""
import os
import sys

import numpy as np
import pytest

from msf_loam_amd import synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
import replay_synthetic as rp  # noqa: E402

pytestmark = pytest.mark.gpu


class OracleBackend:
    def __init__(self, orc):
        self.o = orc

    def extract(self, pts, ring):
        return self.o.extract_features(pts, ring)

    def voxel(self, pts, leaf):
        return self.o.voxel_grid(pts, leaf)

    def scan2scan(self, last, cur, pose):
        rc, p, _ = self.o.match_scan2scan(last["full"][last["less_sharp"]], last["ring"][last["less_sharp"]],
                                          last["full"][last["less_flat"]], last["ring"][last["less_flat"]],
                                          cur["full"][cur["sharp"]], cur["full"][cur["flat"]], pose)
        return p

    def scan2map(self, mc, ms, corner, surf, pose):
        rc, p, _ = self.o.match_scan2map(mc, ms, corner, surf, pose)
        return p

    def new_grids(self):
        return self.o.HybridGrid(3.0, 0.2), self.o.HybridGrid(3.0, 0.4)


def test_replay_matches_oracle_pipeline(oracle):
    world = synth.World(ground_half=45.0)
    truth = rp.trajectory(24)
    est_g, ms = rp.run(rp.GpuBackend(0), world, truth)
    est_o, _ = rp.run(OracleBackend(oracle), world, truth)
    d = np.array([synth.pose_error(a, b) for a, b in zip(est_g, est_o)])
    assert d[:, 0].max() < 1e-4 and d[:, 1].max() < 1e-4, d.max(axis=0)      # north-star tolerance, per scan, chained
    assert d[:, 0].max() < 1e-6, d[:, 0].max()
    # and the SLAM loop actually tracks: bounded absolute error against ground truth
    assert rp.ate(est_g, truth) < 0.3       # cold start: the first scans only have odometry, the map is still empty
    assert all(v < 100.0 for v in ms.values()), ms        # the reference's 100 ms real-time budget per stage


@pytest.mark.gpu
def test_two_handles_on_two_threads(oracle):
    """SURVEY.md §8b threading: the reference runs its odometry and mapping matchers concurrently, each on its
    own thread.  Two handles (two streams, separate scratch) driven from two Python threads (ctypes releases the
    GIL inside the calls) must give exactly what they give when run one after the other."""
    import threading
    from msf_loam_amd import capi
    from tests import common
    _, mc, ms = common.small_world()
    items = []
    for pts, ring, truth, guess in common.scans(4):
        _, corner, surf = common.features_from_oracle(oracle, pts, ring)
        items.append((corner, surf, guess))
    feats = [oracle.extract_features(p, r) for p, r, _, _ in common.scans(4)]
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    def pair(i):
        a, b = feats[i], feats[(i + 1) % 4]
        return (a["full"][a["less_sharp"]], a["ring"][a["less_sharp"]], a["full"][a["less_flat"]], a["ring"][a["less_flat"]],
                b["full"][b["sharp"]], b["full"][b["flat"]], ident)
    mapper, odom = capi.Handle(0), capi.Handle(0)
    mapper.set_map(mc, ms)
    ref_map = [mapper.match_scan2map(*it)[1] for it in items]
    ref_odo = [odom.match_scan2scan(*pair(i))[1] for i in range(4)]
    out_map, out_odo, errs = [], [], []
    def run_map():
        try:
            for rep in range(5):
                out_map.append([mapper.match_scan2map(*it)[1] for it in items])
        except Exception as e:          # surfaced below: an exception inside a thread would otherwise be lost
            errs.append(e)
    def run_odo():
        try:
            for rep in range(5):
                out_odo.append([odom.match_scan2scan(*pair(i))[1] for i in range(4)])
        except Exception as e:
            errs.append(e)
    ta, tb = threading.Thread(target=run_map), threading.Thread(target=run_odo)
    ta.start(); tb.start(); ta.join(); tb.join()
    assert not errs, errs
    for rep in range(5):
        assert all(np.array_equal(a, b) for a, b in zip(out_map[rep], ref_map))
        assert all(np.array_equal(a, b) for a, b in zip(out_odo[rep], ref_odo))
    mapper.close(); odom.close()
