"""N4: ROS-free KITTI reader, pose log in proto/msg.proto's wire format, ATE."""
import os

import numpy as np
import pytest

from msf_loam_amd import dataset, synth


def _pbdata_class():
    """proto/msg.proto:1-37 rebuilt as a dynamic descriptor: an independent decoder for our bytes."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="msg.proto", package="proto", syntax="proto3")
    D, F = descriptor_pb2.FieldDescriptorProto, None

    def msg(name, fields):
        m = fd.message_type.add(name=name)
        for i, (fname, ftype, tname, rep) in enumerate(fields, 1):
            f = m.field.add(name=fname, number=i, type=ftype, label=D.LABEL_REPEATED if rep else D.LABEL_OPTIONAL)
            if tname:
                f.type_name = ".proto." + tname
    msg("Vector3d", [(c, D.TYPE_DOUBLE, None, False) for c in "xyz"])
    msg("Quaterniond", [(c, D.TYPE_DOUBLE, None, False) for c in "xyzw"])
    msg("Rigid3d", [("translation", D.TYPE_MESSAGE, "Vector3d", False), ("rotation", D.TYPE_MESSAGE, "Quaterniond", False)])
    msg("ImuData", [("timestamp", D.TYPE_UINT64, None, False), ("linear_acceleration", D.TYPE_MESSAGE, "Vector3d", False),
                    ("angular_velocity", D.TYPE_MESSAGE, "Vector3d", False)])
    msg("OdometryData", [("timestamp", D.TYPE_UINT64, None, False), ("pose", D.TYPE_MESSAGE, "Rigid3d", False)])
    msg("PbData", [("imu_datas", D.TYPE_MESSAGE, "ImuData", True), ("odom_datas", D.TYPE_MESSAGE, "OdometryData", True)])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("proto.PbData"))


def test_pose_log_is_protobuf_wire_compatible():
    log = dataset.PoseLog()
    poses = synth.random_poses(5, 12)
    poses[2, 0] = 0.0                                           # a zero field is omitted on the wire
    for i, p in enumerate(poses):
        log.add_odom(dataset.from_seconds(1.6e9 + 0.1 * i), p)
    log.add_imu(dataset.from_seconds(1.6e9), [0.1, -9.8, 0.0], [0.01, 0.0, -0.02])
    data = log.serialize()
    PbData = _pbdata_class()
    pb = PbData()
    pb.ParseFromString(data)
    assert len(pb.odom_datas) == 5 and len(pb.imu_datas) == 1
    for i, p in enumerate(poses):
        o = pb.odom_datas[i]
        assert o.timestamp == dataset.from_seconds(1.6e9 + 0.1 * i)
        got = [o.pose.translation.x, o.pose.translation.y, o.pose.translation.z,
               o.pose.rotation.x, o.pose.rotation.y, o.pose.rotation.z, o.pose.rotation.w]
        assert got == list(p)
    assert pb.imu_datas[0].linear_acceleration.y == -9.8 and pb.imu_datas[0].angular_velocity.z == -0.02
    assert pb.SerializeToString(deterministic=True) == data     # byte-identical to protobuf's own encoder
    back = dataset.PoseLog.parse(pb.SerializeToString())
    assert len(back.odom) == 5 and all(np.array_equal(a[1], b[1]) and a[0] == b[0] for a, b in zip(back.odom, log.odom))
    assert back.imu[0][0] == log.imu[0][0] and np.array_equal(back.imu[0][1], log.imu[0][1])


def test_time_ticks_are_nanoseconds_truncated():
    assert dataset.from_seconds(1.5) == 1_500_000_000
    assert dataset.from_seconds(-0.0000000015) == -1            # duration_cast truncates toward zero
    assert dataset.to_seconds(2_500_000_000) == 2.5


def test_kitti_layout_round_trip(tmp_path):
    w = synth.World(ground_half=20.0)
    poses = synth.random_poses(3, 21)
    scans = []
    for i, p in enumerate(poses):
        pts, ring = synth.make_scan(w, p, 500 + i, n_beams=64, n_az=400, elev=(-24.8, 2.0))
        pts[:, 3] = 0.5                                         # KITTI's 4th float is reflectance
        scans.append(pts)
    c, s = np.cos(0.3), np.sin(0.3)
    Tr = (np.array([[0.0, -1, 0], [0, 0, -1], [1, 0, 0]]) @ np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]]), np.array([0.1, -0.2, 0.3]))
    dataset.write_kitti_sequence(str(tmp_path), "07", scans, [0.0, 0.1, 0.2], poses_lidar=poses, Tr=Tr)
    seq = dataset.KittiSequence(str(tmp_path), "07")
    assert len(seq) == 3 and np.allclose(seq.times, [0, 0.1, 0.2])
    assert np.allclose(seq.Tr[0], Tr[0], atol=1e-6) and np.allclose(seq.Tr[1], Tr[1], atol=1e-6)
    for i in range(3):
        dt, dr = synth.pose_error(seq.ground_truth[i], poses[i])
        assert dt < 1e-4 and dr < 1e-5                          # f32 parse (stof) of the text file
        assert abs(np.linalg.norm(seq.ground_truth[i][3:]) - 1) < 1e-12
    raw = dataset.read_kitti_bin(os.path.join(str(tmp_path), "sequences", "07", "velodyne", "000001.bin"))
    assert np.array_equal(raw, scans[1])
    pts, ring = seq.scan(1)
    # the synthetic beams sit on the bin centres: every point gets its beam index back
    _, true_ring = synth.make_scan(w, poses[1], 501, n_beams=64, n_az=400, elev=(-24.8, 2.0))
    assert len(pts) == len(scans[1]) and np.array_equal(ring, true_ring.astype(np.uint16))
    assert np.all(pts[:, 3] == 0)
    with open(os.path.join(str(tmp_path), "sequences", "07", "calib.txt"), "w") as f:
        f.write("P0: 0\nP1: 0\n")
    with pytest.raises(ValueError):
        dataset.read_kitti_calib_tr(os.path.join(str(tmp_path), "sequences", "07", "calib.txt"))


def test_rings_from_elevation_drops_points_outside_the_fan():
    pts = np.array([[10, 0, 0, 0], [10, 0, 10 * np.tan(np.radians(2.0)), 0], [10, 0, 10 * np.tan(np.radians(-24.8)), 0],
                    [10, 0, 10 * np.tan(np.radians(5.0)), 0], [10, 0, -10, 0]], np.float32)
    ring, keep = dataset.rings_from_elevation(pts)
    assert list(keep) == [True, True, True, False, False]
    assert ring[1] == 63 and ring[2] == 0 and ring[0] == round(24.8 / (26.8 / 63))


def test_ate_is_invariant_to_a_rigid_offset():
    truth = synth.random_poses(40, 3)
    R = synth.quat_to_matrix(synth.quat_from_euler(0.1, -0.2, 1.0))
    est = truth.copy()
    est[:, :3] = truth[:, :3] @ R.T + np.array([5.0, -3.0, 1.0])
    assert dataset.ate_rmse(est, truth) < 1e-9
    assert dataset.ate_rmse(est, truth, align=False) > 1.0
    est[:, 0] += np.where(np.arange(40) % 2 == 0, 0.1, -0.1)
    assert 0.05 < dataset.ate_rmse(est, truth) < 0.11


@pytest.mark.gpu
def test_kitti_layout_replays_through_the_pipeline(gpu, oracle, tmp_path):
    """A short 64-beam sequence written in KITTI layout, read back ring-less, re-ringed, and run through
    extraction + scan-to-scan + scan-to-map on the device; the pose log round-trips the estimates."""
    w = synth.World(ground_half=synth.ground_half_for_target(50000))
    mc, ms = synth.make_map(w)
    base = synth.random_poses(2, 77)[1]
    poses = [base]
    rng = np.random.default_rng(2)
    for _ in range(3):
        poses.append(synth.perturb_pose(poses[-1], rng, 0.2, 1.5))
    scans = [synth.make_scan(w, p, 900 + i, n_beams=64, n_az=1900, elev=(-24.8, 2.0))[0] for i, p in enumerate(poses)]
    dataset.write_kitti_sequence(str(tmp_path), "00", scans, 0.1 * np.arange(4), poses_lidar=np.array(poses))
    seq = dataset.KittiSequence(str(tmp_path), "00")
    gpu.set_map(mc, ms)
    log = dataset.PoseLog()
    est, prev = [], None
    for i in range(len(seq)):
        pts, ring = seq.scan(i)
        f = gpu.extract_features(pts, ring)
        fo = oracle.extract_features(pts, ring)
        assert all(np.array_equal(f[k], fo[k]) for k in ("sharp", "less_sharp", "flat", "less_flat"))
        corner = gpu.voxel_downsample(f["full"][f["less_sharp"]], 0.2)
        surf = gpu.voxel_downsample(f["full"][f["less_flat"]], 0.4)
        guess = synth.perturb_pose(seq.ground_truth[i], rng, 0.1, 1.0)
        s, pose, _ = gpu.match_scan2map(corner, surf, guess)
        assert s == 0
        dt, dr = synth.pose_error(pose, poses[i])
        assert dt < 0.05 and dr < 0.01
        if prev is not None:
            s, rel, _ = gpu.match_scan2scan(prev["full"][prev["less_sharp"]], prev["ring"][prev["less_sharp"]],
                                            prev["full"][prev["less_flat"]], prev["ring"][prev["less_flat"]],
                                            f["full"][f["sharp"]], f["full"][f["flat"]], np.array([0, 0, 0, 0, 0, 0, 1.0]))
            assert s == 0
        prev = f
        est.append(pose)
        log.add_odom(dataset.from_seconds(float(seq.times[i])), pose)
    assert dataset.ate_rmse(est, seq.ground_truth, align=False) < 0.05
    path = os.path.join(str(tmp_path), "odom.pb")
    log.save(path)
    back = dataset.PoseLog.load(path)
    assert all(np.array_equal(a[1], b) for a, b in zip(back.odom, est))


@pytest.mark.gpu
@pytest.mark.parametrize("beams", [16, 64])
def test_kitti_layout_replays_through_the_device_resident_slam_step(oracle, tmp_path, beams):
    """A sequence written in KITTI layout (velodyne/*.bin without ring ids, times.txt, poses), read back through
    dataset.KittiSequence (rings re-derived from elevation) and fed to msfl_slam_add_scan scan by scan: the poses must
    equal the oracle-driven odometry + mapping loop on the same clouds.  64 beams: a ~100 k-point less-flat list is beyond
    what the one-workgroup voxel filters address (65 535 points), so this also covers the point-sort voxel form, and the several-
    workgroups forms of the ring split, the list compaction and the scan-to-scan index build that a one-scan call takes."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import replay_synthetic as rp
    from tests.test_gpu_replay import OracleBackendRigid3d
    n = 12 if beams == 16 else 5
    world = synth.World(ground_half=45.0)
    truth = rp.trajectory(300)[:n]
    kw = dict(n_beams=64, n_az=1900, elev=(-24.8, 2.0)) if beams == 64 else {}
    clouds = [synth.make_scan(world, truth[k], synth.SEED + 7000 + k, **kw)[0] for k in range(n)]
    dataset.write_kitti_sequence(str(tmp_path), "07", clouds, 0.1 * np.arange(n), poses_lidar=truth)
    seq = dataset.KittiSequence(str(tmp_path), "07")
    ring_kw = dict(n_scans=64, low_deg=-24.8, high_deg=2.0) if beams == 64 else dict(n_scans=16, low_deg=-15.0, high_deg=15.0)
    scans = [seq.scan(i, **ring_kw) for i in range(len(seq))]
    assert len(scans) == n and all(int(r.max()) == beams - 1 for _, r in scans)
    est_o, _ = rp.run(OracleBackendRigid3d(oracle), world, truth, scans=scans)
    for pipelined in (False, True):
        est_g, recs, _ = rp.run_slam(world, truth, pipelined=pipelined, scans=scans)
        assert all(r.status_extract == 0 for r in recs) and all(r.status_mapping == 0 for r in recs[1:]), [(r.status_extract, r.status_mapping) for r in recs]
        d = np.array([synth.pose_error(a, b) for a, b in zip(est_g, est_o)])
        assert d[:, 0].max() < 1e-6 and d[:, 1].max() < 1e-6, (beams, pipelined, d.max(axis=0))
    assert dataset.ate_rmse(est_g, seq.ground_truth, align=False) < 0.3
