"""The device-resident per-scan SLAM step (msfl_slam_*) at its edges: what the reference CHECK-aborts on or leaves to its
callers becomes a status in the record, and the pipeline keeps running.  Parity of the whole chain against the oracle-driven
loop is in tests/test_gpu_replay.py (300 scans) and tests/test_dataset_io.py (KITTI layout, 16 / 64 beams)."""
import numpy as np
import pytest

from msf_loam_amd import synth
from tests import common

pytestmark = pytest.mark.gpu


def _scans(n):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import replay_synthetic as rp
    world = synth.World(ground_half=45.0)
    truth = rp.trajectory(300)[:n]
    return truth, [synth.make_scan(world, truth[k], synth.SEED + 5000 + k) for k in range(n)]


def test_first_scan_only_initialises_and_the_gate_is_reported():
    """laser_odometry.cc:72-73 (first scan: no MatchScan2Scan) and laser_mapping.cc:284-285 (empty map: MatchScan2Map skipped,
    the pose guess passes through, the scan is still inserted)."""
    from msf_loam_amd import capi
    truth, scans = _scans(3)
    s = capi.Slam(0, max_scan_points=max(len(p) for p, _ in scans), max_rings=16, pose_odom2map=truth[0])
    r0 = s.add_scan(*scans[0])
    assert r0.scan_index == 0 and r0.status_extract == 0 and r0.status_mapping == capi.MAP_TOO_SMALL
    assert r0.n_map_corner == 0 and r0.n_map_surf == 0 and list(r0.mapping.lm_iterations) == [0, 0] and r0.odometry.status == 0
    assert np.array_equal(np.array(r0.pose_odom[:]), [0, 0, 0, 0, 0, 0, 1.0])             # pose_scan2world_ starts at identity
    assert max(synth.pose_error(np.array(r0.pose_map[:]), truth[0])) < 1e-12              # odom2map * identity, unrefined
    assert r0.grid_corner[0] > 100 and r0.grid_surf[0] > 5000 and r0.grid_corner[3] == 0  # inserted all the same (:330-338)
    r1 = s.add_scan(*scans[1])
    assert r1.status_mapping == 0 and 10 < r1.n_map_corner <= r0.grid_corner[0] and 50 < r1.n_map_surf <= r0.grid_surf[0]   # whole cells around the scan
    assert sum(r1.mapping.lm_iterations) > 0 and sum(r1.odometry.lm_iterations) > 0
    assert r1.n_corner_ds <= r1.n_less_sharp and r1.n_surf_ds <= r1.n_less_flat and r1.n_sharp <= 2 * 6 * 16 and r1.n_flat <= 4 * 6 * 16
    gc, gs = s.grids()
    assert gc.size()[0] == r1.grid_corner[0] and gs.size() == (r1.grid_surf[0], r1.grid_surf[1])
    assert len(gs.dump()) == r1.grid_surf[0]
    s.close()


def test_device_resident_scans_equal_host_scans_bit_for_bit():
    import torch
    from msf_loam_amd import capi
    truth, scans = _scans(6)
    cap = max(len(p) for p, _ in scans)
    a = capi.Slam(0, max_scan_points=cap, max_rings=16, pose_odom2map=truth[0])
    b = capi.Slam(0, max_scan_points=cap, max_rings=16, pose_odom2map=truth[0])
    dev = torch.device("cuda", 0)
    keep = []
    for k, (pts, ring) in enumerate(scans):
        ra = a.add_scan(pts, ring)
        d_pts, d_ring = torch.from_numpy(pts).to(dev), torch.from_numpy(ring.astype(np.int16)).to(dev)
        keep.append((d_pts, d_ring))                      # device inputs must stay untouched until the record is out
        torch.cuda.synchronize()
        rb = b.add_scan_device(d_pts.data_ptr(), d_ring.data_ptr(), len(pts))
        assert np.array_equal(np.array(ra.pose_map[:]), np.array(rb.pose_map[:])), k
        assert list(ra.grid_surf)[:3] == list(rb.grid_surf)[:3]
    a.close(); b.close()


def test_bad_scans_become_statuses_and_the_pipeline_goes_on():
    """ring >= 128 (CHECK_LT, msf_loam_node.cc:136), no valid point (CHECK_GT :186), a ring id beyond the configured ring
    count (the launches are sized for max_rings), more points than the configured capacity, a record that is no longer held."""
    from msf_loam_amd import capi
    truth, scans = _scans(5)
    cap = max(len(p) for p, _ in scans)
    s = capi.Slam(0, max_scan_points=cap, max_rings=16, pose_odom2map=truth[0])
    good = [s.add_scan(*scans[0]), s.add_scan(*scans[1])]
    pts, ring = scans[2]
    bad_ring = ring.copy(); bad_ring[100] = 200
    r = s.add_scan(pts, bad_ring)
    # a scan that is not matched at all is not a successful MatchScan2Scan: the odometry record says so too
    assert r.status_extract == capi.BAD_RING and r.status_mapping == capi.BAD_ARG and r.odometry.status == capi.BAD_ARG
    assert list(r.odometry.lm_iterations) == [0, 0] and r.status_imu == 0 and r.status_insert == 0
    assert np.array_equal(np.array(r.pose_odom[:]), np.array(good[1].pose_odom[:]))        # the chain is left where it was
    assert list(r.grid_surf)[:2] == list(good[1].grid_surf)[:2]                            # nothing inserted
    nan = pts.copy(); nan[:, :3] = np.nan
    r = s.add_scan(nan, ring)
    assert r.status_extract == capi.BAD_ARG
    # the pipeline recovers: the next good scan is matched against the last GOOD scan's features?  No — like the reference,
    # scan_last_ is whatever came last; after a scan without features the odometry reports too few correspondences and keeps its pose
    r = s.add_scan(*scans[3])
    assert r.status_extract == 0 and r.odometry.status == capi.TOO_FEW_CORRESPONDENCES and r.status_mapping == 0
    r = s.add_scan(*scans[4])
    assert r.status_extract == 0 and r.odometry.status == 0 and r.status_mapping == 0
    assert max(synth.pose_error(np.array(r.pose_map[:]), truth[4])) < 0.2                 # mapping pulls the pose back (one odometry step was lost)
    with pytest.raises(capi.MsflError) as e:
        s.add_scan(np.concatenate([pts, pts]), np.concatenate([ring, ring]))
    assert e.value.status == capi.CAPACITY
    with pytest.raises(capi.MsflError) as e:
        s.result(0)
    assert e.value.status == capi.BAD_ARG
    s.close()
    # a sensor with more rings than configured: reported, not matched, no out-of-bounds launch
    s = capi.Slam(0, max_scan_points=cap, max_rings=4, pose_odom2map=truth[0])
    r = s.add_scan(*scans[0])
    assert r.status_extract == capi.CAPACITY and r.status_mapping == capi.BAD_ARG and r.n_sharp > 2 * 6 * 4
    s.close()


def test_map_store_compaction_under_the_slam_step_keeps_the_poses():
    """The point pool is compacted whenever the host's worst-case bound says the next insert might not fit; with a pool this
    small that happens every few scans.  Poses must not depend on it: run the same scans with the default pool and with
    MSFL_GRID_MIN_POOL forced tiny."""
    import os
    from msf_loam_amd import capi
    truth, scans = _scans(40)
    cap = max(len(p) for p, _ in scans)

    def run():
        s = capi.Slam(0, max_scan_points=cap, max_rings=16, pose_odom2map=truth[0])
        out = [np.array(s.add_scan(*sc).pose_map[:]) for sc in scans]
        s.close()
        return np.array(out)
    ref = run()
    os.environ["MSFL_GRID_MIN_POOL"] = "65536"
    try:
        small = run()
    finally:
        del os.environ["MSFL_GRID_MIN_POOL"]
    assert np.array_equal(ref, small)


def test_launch_bounds_do_not_change_the_result():
    """Every launch of the step is sized by host-known upper bounds (scan capacity, pick limits per ring count) and reads the
    real sizes on the device; short lists take one-workgroup forms of the map store's sort / touch-list kernels, long ones the
    device-wide ones.  A pipeline configured for 128 rings and 200 000 points per scan (the defaults: corner lists of up to
    15 360 points, every launch several times larger) must give the poses of one configured tightly, bit for bit."""
    from msf_loam_amd import capi
    truth, scans = _scans(12)
    cap = max(len(p) for p, _ in scans)
    tight = capi.Slam(0, max_scan_points=cap, max_rings=16, pose_odom2map=truth[0])
    loose = capi.Slam(0, max_scan_points=200000, max_rings=128, pose_odom2map=truth[0])
    for k, sc in enumerate(scans):
        a, b = tight.add_scan(*sc), loose.add_scan(*sc)
        assert np.array_equal(np.array(a.pose_map[:]), np.array(b.pose_map[:])), k
        assert np.array_equal(np.array(a.pose_odom[:]), np.array(b.pose_odom[:])), k
        assert (a.n_corner_ds, a.n_surf_ds, a.n_map_corner, a.n_map_surf) == (b.n_corner_ds, b.n_surf_ds, b.n_map_corner, b.n_map_surf)
        assert list(a.grid_corner)[:3] == list(b.grid_corner)[:3] and list(a.grid_surf)[:3] == list(b.grid_surf)[:3]
    tight.close(); loose.close()


def test_mapping_thread_gives_the_same_poses(monkeypatch):
    """MSFL_SLAM_THREADS=1: the mapping chain of every scan is enqueued by a second host thread (the reference's own layout,
    laser_mapping.cc:86).  The GPU-side order is fixed by events, so poses, counts and map stores equal the single-thread
    form bit for bit, pipelined and synchronous."""
    import os
    sys_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples")
    import sys
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    import replay_synthetic as rp
    world = synth.World(ground_half=45.0)
    truth = rp.trajectory(120)[:40]
    scans = [synth.make_scan(world, truth[k], synth.SEED + 5000 + k) for k in range(len(truth))]
    monkeypatch.setenv("MSFL_SLAM_THREADS", "0")
    ref, recs0, _ = rp.run_slam(world, truth, pipelined=True, scans=scans)
    monkeypatch.setenv("MSFL_SLAM_THREADS", "1")
    for pipelined in (True, False):
        maps = {}
        est, recs, _ = rp.run_slam(world, truth, pipelined=pipelined, scans=scans, maps_out=maps)
        assert np.array_equal(est, ref)
        assert [list(r.grid_surf)[:2] for r in recs] == [list(r.grid_surf)[:2] for r in recs0]


@pytest.mark.parametrize("threads", ["0", "1"])
def test_a_poisoned_pipeline_still_hands_out_the_earlier_results(monkeypatch, threads):
    """a failure in the middle of scan k's chain poisons the pipeline (every later msfl_slam_add_scan returns the stored
    error) but the records of the scans whose chains were completely enqueued BEFORE it stay fetchable, as msfl_slam_add_scan_imu's
    contract says; msfl_slam_grids still works.  The failure is injected (MSFL_SLAM_FAIL_AT): nothing in a healthy run produces one."""
    from msf_loam_amd import capi
    truth, scans = _scans(6)
    ref = capi.Slam(0, max_scan_points=max(len(p) for p, _ in scans), max_rings=16, pose_odom2map=truth[0])
    want = [np.array(ref.add_scan(*scans[k]).pose_map[:]) for k in range(3)]
    ref.close()
    monkeypatch.setenv("MSFL_SLAM_THREADS", threads)
    monkeypatch.setenv("MSFL_SLAM_FAIL_AT", "3")
    slam = capi.Slam(0, max_scan_points=max(len(p) for p, _ in scans), max_rings=16, pose_odom2map=truth[0])
    try:
        for k in range(3):
            slam.add_scan(*scans[k], wait=False)
        failed = False
        try:
            slam.add_scan(*scans[3], wait=False)        # single-thread form: fails here; mapping thread: the failure surfaces at the next wait
        except capi.MsflError as e:
            failed = True
            assert e.status == capi.HIP_ERROR and "injected" in str(e)
        for k in (1, 2):                                # fetched AFTER the failure
            assert np.array_equal(np.array(slam.result(k).pose_map[:]), want[k]), k
        with pytest.raises(capi.MsflError) as e3:       # the failed scan itself has no result
            slam.result(3)
        assert e3.value.status == capi.HIP_ERROR
        with pytest.raises(capi.MsflError) as e4:       # and nothing follows it
            slam.add_scan(*scans[4], wait=False)
        assert e4.value.status == capi.HIP_ERROR
        assert failed or threads == "1"
        gc_, gs_ = slam.grids()                         # the stores are still readable (what scans 0-2 inserted)
        assert gs_.size()[0] > 1000
    finally:
        slam.close()


def test_concurrent_sessions_in_one_process_do_not_disturb_each_other():
    """Several msfl_slam objects driven from their own host threads (ctypes releases the GIL inside the calls) share the process's HIP
    runtime and nothing else: every session's pose track must equal the single session's bit for bit, pipelined use included
    (tools/slam_sessions.py measures what such replicas are worth: nothing, profiles/r05b_slam_sessions.md)."""
    import sys, os, threading
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import replay_synthetic as rp
    truth, scans = _scans(25)
    world = synth.World(ground_half=45.0)
    ref, _, _ = rp.run_slam(world, truth, pipelined=False, scans=scans)
    K = 4
    est = [None] * K
    gate = threading.Barrier(K)

    def work(i):
        gate.wait()
        est[i], _, _ = rp.run_slam(world, truth, pipelined=bool(i % 2), scans=scans)

    th = [threading.Thread(target=work, args=(i,)) for i in range(K)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert all(e is not None and np.array_equal(e, ref) for e in est)


def test_batches_beyond_the_2d_launch_limit_are_refused_up_front(gpu):
    """kernels launched 2-D over (tile, scan / pair) cannot take more than 65 535 rows: MSFL_CAPACITY with a message, before
    any staging, instead of an opaque HIP launch error."""
    import ctypes as C
    from msf_loam_amd import capi
    P = 65536
    z = np.zeros(P + 1, np.int32)
    poses = np.zeros((P, 7)); status = np.zeros(P, np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    s = gpu.lib.msfl_match_pairs_batch(gpu.h, C.c_int(P), None, vp(z), None, vp(z), None, vp(z), None, vp(z), vp(poses), vp(status), None, C.c_int(capi.MEM_HOST))
    assert s == capi.CAPACITY and b"65535" in gpu.lib.msfl_last_error(gpu.h)
    fb = capi.FeaturesBatch()
    s = gpu.lib.msfl_extract_features_batch(gpu.h, C.c_int(P), None, None, vp(z), C.byref(fb), vp(status), C.c_int(capi.MEM_HOST))
    assert s == capi.CAPACITY and b"65535" in gpu.lib.msfl_last_error(gpu.h)
