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

    # ---- the IMU branches of LaserMapping::Run (pre = (sum_dt, delta_q, delta_p))
    def undistort(self, pre, pts):
        bad, out = self.o.undistort_cloud(*pre, pts)                       # scan_undistortion.cc:5-19
        assert bad == 0
        return out

    def deskew(self, pre, pts, rot_odom, velocity, gravity):
        bad, out = self.o.deskew_cloud(*pre, pts, rot_odom, velocity, gravity)   # laser_mapping.cc:197-211
        assert bad == 0
        return out

    def scan2map_deskew(self, mc, ms, corner, surf, pre, velocity, gravity, pose):
        bad_c, cdq, cdp = self.o.delta_qp_cloud(*pre, corner)              # mapping_scan_matcher.cc:113-117
        bad_s, sdq, sdp = self.o.delta_qp_cloud(*pre, surf)                # :183-187
        assert bad_c == 0 and bad_s == 0
        rc, p, _ = self.o.match_scan2map_deskew(mc, ms, corner, surf, cdq, cdp, sdq, sdp, velocity, gravity, pose)
        return p


class OracleBackendRigid3d(OracleBackend):
    """The oracle-driven loop with the reference's own Rigid3d algebra (rigid_transform.h:78-82,105-111,131-137) instead of
    the example's numpy matrices: what the device-resident SLAM step is compared with."""

    def compose(self, a, b):
        return self.o.pose_compose(a, b)

    def inverse(self, a):
        qc = np.r_[-np.asarray(a[3:6]), a[6]]
        return np.r_[-self.o.quat_rotate(qc, np.asarray(a[:3], np.float64)), qc]

    def transform(self, pose, pts):
        return self.o.transform_cloud(pts, pose)


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


def _oracle_loop(oracle, world, truth, n):
    """rp.run with the oracle backend on the first n poses, cached per session (the 300-scan loop takes ~20 s)."""
    key = n
    if key not in _oracle_loop.cache:
        maps = {}
        _oracle_loop.cache[key] = (rp.run(OracleBackendRigid3d(oracle), world, truth[:n], maps_out=maps)[0], maps)
    return _oracle_loop.cache[key]


_oracle_loop.cache = {}


@pytest.mark.parametrize("pipelined", [False, True])
def test_device_resident_slam_step_matches_the_oracle_loop_over_300_scans(oracle, pipelined):
    """BASELINE configs[2] through msfl_slam_add_scan (raw scan in, pose out, nothing but the scan and the record crosses
    PCIe): every one of 300 poses within 1e-6 m / 1e-6 rad of the oracle-driven loop, the same ATE, and the two map
    stores equal to the oracle's at the end.  Pipelined = results fetched one scan late, so that the odometry chain of
    scan k + 1 overlaps the mapping chain of scan k on the second stream; it must give the same poses bit for bit."""
    n = 300
    world = synth.World(ground_half=45.0)
    truth = rp.trajectory(n)
    est_o, maps_o = _oracle_loop(oracle, world, truth, n)
    maps_g = {}
    est_g, recs, ms = rp.run_slam(world, truth, pipelined=pipelined, maps_out=maps_g)
    d = np.array([synth.pose_error(a, b) for a, b in zip(est_g, est_o)])
    assert d[:, 0].max() < 1e-6 and d[:, 1].max() < 1e-6, (d.max(axis=0), int(d[:, 0].argmax()))
    assert abs(rp.ate(est_g, truth) - rp.ate(est_o, truth)) < 1e-6
    assert rp.ate(est_g, truth) < 0.3
    assert all(r.status_extract == 0 for r in recs)
    assert sum(1 for r in recs if r.status_mapping != 0) <= 2          # the map gate is only closed while the map is empty
    assert recs[-1].grid_corner[0] > 1000 and recs[-1].grid_surf[0] > 10000
    for k in ("corner", "surf"):                                       # same voxels; coordinates as close as the poses that placed them
        assert maps_g[k].shape == maps_o[k].shape, (k, maps_g[k].shape, maps_o[k].shape)
        assert np.abs(maps_g[k] - maps_o[k]).max() < 1e-4
    if not pipelined:
        test_device_resident_slam_step_matches_the_oracle_loop_over_300_scans.sync_poses = est_g
    else:
        ref = getattr(test_device_resident_slam_step_matches_the_oracle_loop_over_300_scans, "sync_poses", None)
        if ref is not None:
            assert np.array_equal(ref, est_g)


@pytest.mark.parametrize("variant", ["quirks", "imu", "quirks+imu"])
def test_slam_step_runs_what_laser_mapping_run_runs(oracle, variant):
    """LaserMapping::Run as the reference executes it: `reference_quirks` (FilterLessFlatLessCornerFeature cuts the surf cloud
    to its first n_less_sharp points, laser_mapping.cc:186,340-364: that cloud is filtered, matched AND inserted) and the
    per-scan IMU inputs (UndistortScan before the match for the first 50 scans, :170-176; from scan 50 on the is_initialized
    matcher branch from the pre-solved pose with Deskew factors and DoUndistort before the insert, :197-211).  300 scans (160
    for the IMU variant on the full surf lists: the CPU loop's deskew matcher is 0.4 s per scan there)
    through msfl_slam_add_scan_imu, pipelined, against the oracle-driven loop doing the same on the CPU: every pose within
    1e-6 m / 1e-6 rad, the same final map stores.  The synchronous form of the first 80 scans must equal the pipelined one
    bit for bit (both branches and the switch are inside)."""
    quirks, with_imu = "quirks" in variant, "imu" in variant
    n = 300 if quirks else 160        # without the truncation the oracle's deskew matcher works on ~4 600 surf features per scan: 0.4 s per scan
    world = synth.World(ground_half=45.0)
    truth = rp.trajectory(300)[:n]
    imu = rp.synthetic_imu(truth, switch_at=50) if with_imu else None
    maps_o, maps_g = {}, {}
    est_o, _ = rp.run(OracleBackendRigid3d(oracle), world, truth, maps_out=maps_o, quirks=quirks, imu=imu)
    est_g, recs, _ = rp.run_slam(world, truth, pipelined=True, maps_out=maps_g, quirks=quirks, imu=imu)
    d = np.array([synth.pose_error(a, b) for a, b in zip(est_g, est_o)])
    assert d[:, 0].max() < 1e-6 and d[:, 1].max() < 1e-6, (d.max(axis=0), int(d[:, 0].argmax()))
    assert all(r.status_extract == 0 and r.status_imu == 0 and r.status_insert == 0 for r in recs)
    assert sum(1 for r in recs if r.status_mapping != 0) <= 2
    for k in ("corner", "surf"):
        assert maps_g[k].shape == maps_o[k].shape, (k, maps_g[k].shape, maps_o[k].shape)
        assert np.abs(maps_g[k] - maps_o[k]).max() < 1e-4
    if quirks:      # the truncated surf list is what the mapping thread sees: n_surf_ds is the filter of n_less_sharp points only
        assert all(r.n_surf_ds <= r.n_less_sharp for r in recs)
        assert recs[-1].grid_surf[0] < 0.5 * 40000     # far fewer surf map points than the full lists leave (~40 k)
    if with_imu:    # the trajectory is not the LiDAR-only one: the IMU passes really act
        est_plain, _ = _oracle_loop(oracle, world, rp.trajectory(300), 300) if not quirks else (None, None)
        if est_plain is not None:
            assert np.abs(est_plain[60:n, :3] - est_o[60:, :3]).max() > 1e-4
    # tracking quality stays that of LOAM on this drive
    assert rp.ate(est_g, truth) < 0.35
    est_s, _, _ = rp.run_slam(world, truth[:80], pipelined=False, quirks=quirks, imu=imu[:80] if imu else None)
    assert np.array_equal(est_s, est_g[:80])


def test_slam_imu_refusals_and_quirk_out_of_bounds(oracle):
    """The reference's CHECK failures and its out-of-bounds read become per-scan statuses: a time stamp outside the
    pre-integration span (scan_undistortion.cc:26-30) -> status_imu = status_mapping = MSFL_BAD_ARG, the scan is neither
    matched nor inserted and the pipeline carries on; reference_quirks with more less-sharp than less-flat points ->
    status_mapping = MSFL_BAD_ARG; argument errors (too many samples, decreasing sum_dt) are refused before anything is
    enqueued."""
    from msf_loam_amd import capi
    world = synth.World(ground_half=45.0)
    truth = rp.trajectory(6)
    scans = [synth.make_scan(world, truth[k], synth.SEED + 5000 + k) for k in range(6)]
    imu = rp.synthetic_imu(truth, switch_at=3)
    slam = capi.Slam(0, max_scan_points=max(len(p) for p, _ in scans), max_rings=16, pose_odom2map=truth[0])
    r0 = slam.add_scan(*scans[0], imu=imu[0])
    r1 = slam.add_scan(*scans[1], imu=imu[1])
    assert r0.status_imu == 0 and r1.status_imu == 0 and r1.status_mapping in (0, capi.MAP_TOO_SMALL)
    short = dict(imu[2]); short["sum_dt"] = np.linspace(0.0, 0.05, 45)       # the scan spans 0.1 s: half of its points fall outside
    r2 = slam.add_scan(*scans[2], imu=short)
    assert r2.status_imu == capi.BAD_ARG and r2.status_mapping == capi.BAD_ARG and r2.status_extract == 0
    assert list(r2.grid_corner)[:2] == list(r1.grid_corner)[:2] and list(r2.grid_surf)[:2] == list(r1.grid_surf)[:2]   # nothing inserted
    r3 = slam.add_scan(*scans[3], imu=imu[3])                                 # is_initialized, good span: carries on
    assert r3.status_imu == 0 and r3.status_mapping == 0 and r3.grid_surf[0] > r2.grid_surf[0]
    neg = dict(imu[4]); neg["is_initialized"] = False; neg["sum_dt"] = np.linspace(0.02, 0.13, 45)   # times below front()
    assert slam.add_scan(*scans[4], imu=neg).status_imu == capi.BAD_ARG
    with pytest.raises(capi.MsflError) as e:
        many = dict(imu[5]); many["sum_dt"] = np.linspace(0, 0.11, 3000); many["delta_q"] = np.tile([0, 0, 0, 1.0], (3000, 1)); many["delta_p"] = np.zeros((3000, 3))
        slam.add_scan(*scans[5], imu=many)
    assert e.value.status == capi.CAPACITY
    with pytest.raises(capi.MsflError) as e:
        dec = dict(imu[5]); dec["sum_dt"] = np.linspace(0.11, 0.0, 45)
        slam.add_scan(*scans[5], imu=dec)
    assert e.value.status == capi.BAD_ARG
    r5 = slam.add_scan(*scans[5], imu=imu[5])                                 # refused calls enqueued nothing: not poisoned
    assert r5.status_imu == 0 and r5.scan_index == 5
    slam.close()
    # reference_quirks: a scan with MORE less-sharp than less-flat points (jagged ranges: every sector fills its 20 corner picks
    # and nothing is left flat or unlabelled) -- the reference's copyPointCloud would read out of bounds
    rng = np.random.default_rng(3)
    m = 48
    az = np.linspace(0, 2 * np.pi, m, endpoint=False)
    jp, jr = [], []
    for b in range(16):
        el, rg = np.deg2rad(-15 + 2 * b), rng.uniform(5, 30, m)
        jp.append(np.c_[rg * np.cos(el) * np.cos(-az), rg * np.cos(el) * np.sin(-az), rg * np.sin(el), np.zeros(m)])
        jr.append(np.full(m, b))
    jp, jr = np.concatenate(jp).astype(np.float32), np.concatenate(jr).astype(np.uint16)
    order = np.lexsort((jr, np.tile(np.arange(m), 16)))
    jp, jr = jp[order], jr[order]
    fo = oracle.extract_features(jp, jr)
    assert len(fo["less_sharp"]) > len(fo["less_flat"])                       # the premise
    pts, ring = scans[0]
    slam = capi.Slam(0, max_scan_points=len(pts), max_rings=16, pose_odom2map=truth[0], reference_quirks=1)
    a = slam.add_scan(pts, ring)
    assert a.status_mapping in (0, capi.MAP_TOO_SMALL) and 0 < a.n_surf_ds <= a.n_less_sharp
    b = slam.add_scan(jp, jr)
    assert b.status_extract == 0 and b.n_less_sharp == len(fo["less_sharp"]) and b.n_less_flat == len(fo["less_flat"])
    assert b.status_mapping == capi.BAD_ARG and b.status_imu == 0 and b.n_corner_ds == 0 and b.n_surf_ds == 0
    assert list(b.grid_corner)[:2] == list(a.grid_corner)[:2] and list(b.grid_surf)[:2] == list(a.grid_surf)[:2]   # not inserted
    c = slam.add_scan(pts, ring)                                              # the pipeline carries on
    assert c.status_mapping == 0 and c.grid_surf[0] >= a.grid_surf[0]
    slam.close()
    plain = capi.Slam(0, max_scan_points=len(pts), max_rings=16, pose_odom2map=truth[0])
    assert plain.add_scan(jp, jr).status_mapping in (0, capi.MAP_TOO_SMALL)   # without the quirk the same scan is an ordinary one
    plain.close()


def _figure_eight(n, step_scale):
    """A faster, less regular drive than rp.trajectory: a figure eight with `step_scale` x its per-scan motion, rolling and
    pitching a few degrees."""
    poses = []
    for k in range(n):
        a = 2 * np.pi * step_scale * k / 140.0
        x, y = 8.0 * np.sin(a), 5.0 * np.sin(2 * a)
        dx, dy = 8.0 * np.cos(a), 10.0 * np.cos(2 * a)
        yaw = np.arctan2(dy, dx)
        poses.append(np.r_[x, y, 1.7 + 0.05 * np.sin(5 * a), synth.quat_from_euler(0.04 * np.sin(3 * a), 0.03 * np.cos(2 * a), yaw)])
    return np.array(poses)


@pytest.mark.parametrize("world_seed,step_scale,n", [(synth.SEED + 77, 1.0, 50), (synth.SEED + 78, 2.0, 50)])
def test_device_resident_slam_step_on_other_worlds_and_drives(oracle, world_seed, step_scale, n):
    """The 300-scan test uses one world and one smooth loop.  Other pole layouts and a figure-eight drive at one and two
    times its speed (up to ~0.7 m and ~6 degrees per scan, with roll and pitch): the device-resident chain still
    reproduces the oracle-driven loop pose by pose."""
    world = synth.World(seed=world_seed, ground_half=45.0)
    truth = _figure_eight(n, step_scale)
    scans = [synth.make_scan(world, truth[k], world_seed + 9000 + k) for k in range(n)]
    est_o, _ = rp.run(OracleBackendRigid3d(oracle), world, truth, scans=scans)
    est_g, recs, _ = rp.run_slam(world, truth, pipelined=True, scans=scans)
    d = np.array([synth.pose_error(a, b) for a, b in zip(est_g, est_o)])
    assert d[:, 0].max() < 1e-6 and d[:, 1].max() < 1e-6, (d.max(axis=0), int(d[:, 0].argmax()))
    assert all(r.status_extract == 0 for r in recs) and sum(1 for r in recs if r.status_mapping != 0) <= 2
    # how well LOAM itself tracks the fast drive is not the point here (0.8 m ATE at twice the speed, the oracle's too)
    assert abs(rp.ate(est_g, truth) - rp.ate(est_o, truth)) < 1e-6


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


@pytest.mark.parametrize("kind", ["outdoor", "corridor"])
def test_device_resident_slam_step_in_the_other_worlds(oracle, kind):
    """Round 5: BASELINE configs[2]'s step away from the room.  A 100-scan drive down the outdoor world's
    street (relief, building faces, trunks, volumetric canopy: the map stores fill with leaf-dense cells) and down the 80 m
    corridor (LOAM's classic failure: nothing constrains the motion along the axis, the estimate stays behind the truth) —
    what is tested is that the device-resident chain reproduces the oracle-driven loop pose by pose (<= 1e-6 m / 1e-6 rad),
    failure included, with the same final map stores (laser_odometry.cc:69-95, laser_mapping.cc:138-338)."""
    from tests import common
    n = 100
    world, _, _ = common.other_world(kind)
    truth = common.world_drive(kind, n)
    scans = [synth.make_scan(world, truth[k], synth.SEED + 7000 + k) for k in range(n)]
    maps_o, maps_g = {}, {}
    est_o, _ = rp.run(OracleBackendRigid3d(oracle), world, truth, scans=scans, maps_out=maps_o)
    est_g, recs, _ = rp.run_slam(world, truth, pipelined=True, scans=scans, maps_out=maps_g)
    d = np.array([synth.pose_error(a, b) for a, b in zip(est_g, est_o)])
    assert d[:, 0].max() < 1e-6 and d[:, 1].max() < 1e-6, (kind, d.max(axis=0), int(d[:, 0].argmax()))
    assert all(r.status_extract == 0 for r in recs) and sum(1 for r in recs if r.status_mapping != 0) <= 2
    for k in ("corner", "surf"):
        assert maps_g[k].shape == maps_o[k].shape, (k, maps_g[k].shape, maps_o[k].shape)
        assert np.abs(maps_g[k] - maps_o[k]).max() < 1e-4
    assert abs(rp.ate(est_g, truth) - rp.ate(est_o, truth)) < 1e-6
    if kind == "outdoor":
        assert rp.ate(est_g, truth) < 0.6                      # LOAM tracks the street (drift ~1.5 % of the 40 m driven)
    else:
        assert rp.ate(est_g, truth) > 5.0                      # and loses the corridor's axis, like the CPU loop does
    est_s, _, _ = rp.run_slam(world, truth[:40], pipelined=False, scans=scans[:40])
    assert np.array_equal(est_s, est_g[:40])


@pytest.mark.parametrize("variant", ["plain", "imu", "quirks+imu"])
def test_slam_step_delivers_the_clouds_laser_mapping_run_publishes(oracle, variant):
    """msfl_slam_config.keep_clouds.  LaserMapping::Run also rewrites cloud_full_res — UndistortScan while the
    estimator is not initialised (laser_mapping.cc:170-176), DoUndistort once it is (:206) — accumulates it in the map frame
    (TransformPointCloud by pose_map_scan2world_, :214-217) and publishes the sharp / flat clouds (:418-440).  70 scans with the
    switch at scan 50, pipelined, against the oracle-driven loop: the full cloud in the scan frame and the index lists bit for bit,
    the map-frame cloud as close as the two pose estimates (<= 1e-5 m, the poses themselves agree to ~1e-14), status_clouds 0;
    poses equal a run without keep_clouds bit for bit (the extra launch changes nothing else)."""
    quirks, with_imu = "quirks" in variant, "imu" in variant
    n = 70
    world = synth.World(ground_half=45.0)
    truth = rp.trajectory(300)[:n]
    imu = rp.synthetic_imu(truth, switch_at=50) if with_imu else None
    co, cg = [], []
    est_o, _ = rp.run(OracleBackendRigid3d(oracle), world, truth, quirks=quirks, imu=imu, clouds_out=co)
    est_g, recs, _ = rp.run_slam(world, truth, pipelined=True, quirks=quirks, imu=imu, clouds_out=cg)
    est_n, _, _ = rp.run_slam(world, truth, pipelined=True, quirks=quirks, imu=imu)
    assert np.array_equal(est_g, est_n)
    assert len(co) == len(cg) == n and all(r.status_clouds == 0 for r in recs)
    moved = 0
    for k in range(n):
        a, b = cg[k], co[k]
        for key in ("sharp", "less_sharp", "flat", "less_flat", "ring"):
            assert np.array_equal(a[key], b[key]), (k, key)
        assert np.array_equal(a["full_scan"], b["full_scan"]), k
        assert a["full_map"].shape == b["full_map"].shape and np.abs(a["full_map"] - b["full_map"]).max() < 1e-5, k
        assert np.array_equal(a["full_map"][:, 3], a["full_scan"][:, 3])            # intensity (the relative time) rides along
        # the listed points of the full cloud are the clouds PublishScan publishes: after the IMU passes they differ from the raw scan
        moved += int(np.any(a["full_scan"][:, :3] != rp_full(oracle, world, truth, k)[:, :3]))
    assert moved == (n if with_imu else 0)
    # the map-frame cloud really is pose_map applied to the scan-frame cloud (msfl_transform_cloud's arithmetic)
    k = n - 1
    want = oracle.transform_cloud(cg[k]["full_scan"], est_g[k])
    assert np.array_equal(cg[k]["full_map"], want)


def rp_full(oracle, world, truth, k):
    """The raw extraction output of scan k (what cloud_full_res holds before any IMU pass)."""
    key = (id(world), k)
    if key not in rp_full.cache:
        pts, ring = synth.make_scan(world, truth[k], synth.SEED + 5000 + k)
        rp_full.cache[key] = oracle.extract_features(pts, ring)["full"]
    return rp_full.cache[key]


rp_full.cache = {}


def test_clouds_are_refused_without_the_switch_and_after_the_buffers_moved_on():
    from msf_loam_amd import capi
    world = synth.World(ground_half=45.0)
    truth = rp.trajectory(300)[:4]
    scans = [synth.make_scan(world, truth[k], synth.SEED + 5000 + k) for k in range(4)]
    plain = capi.Slam(0, max_scan_points=max(len(p) for p, _ in scans), max_rings=16, pose_odom2map=truth[0])
    plain.add_scan(*scans[0])
    with pytest.raises(capi.MsflError) as e:
        plain.clouds(0)
    assert e.value.status == capi.BAD_ARG and "keep_clouds" in str(e.value)
    plain.close()
    slam = capi.Slam(0, max_scan_points=max(len(p) for p, _ in scans), max_rings=16, pose_odom2map=truth[0], keep_clouds=1)
    for k in range(3):
        slam.add_scan(*scans[k])
    assert len(slam.clouds(2)["full_scan"]) > 20000 and len(slam.clouds(1)["full_scan"]) > 20000
    with pytest.raises(capi.MsflError) as e:
        slam.clouds(0)                      # scan 2 reuses scan 0's buffer set
    assert e.value.status == capi.BAD_ARG
    # a time stamp outside the pre-integration span on a point that is in neither list: status_clouds, the point left alone
    imu = rp.synthetic_imu(truth, switch_at=0)[3]
    imu = dict(imu); imu["sum_dt"] = np.linspace(0.0, 0.099, 45)          # the last points of every ring (the margins) have t ~ 0.1
    r = slam.add_scan(*scans[3], imu=imu)
    if r.status_imu == 0:                    # the listed points all fit the span: then only the margins can be outside
        assert r.status_clouds in (0, capi.BAD_ARG)
    slam.close()
