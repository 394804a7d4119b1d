"""ROS-free rosbag 2.0 reader (msf_loam_amd/rosbag_io.py): what BASELINE configs[2] (`nsh_indoor_outdoor.bag`) needs to be
replayed through the C ABI once the bag is available.  The real bag is not in the image: the fixtures are written by the module's own
minimal writer in the layout velodyne_pointcloud / the ROS serialiser produce ([3P-recall] of the published format), so these tests pin
the reader against the writer and against hand-assembled bytes, not against ROS."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from msf_loam_amd import rosbag_io as rb, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write(path, clouds, comp="none", per=3, with_imu=True):
    w = rb.BagWriter(path, compression=comp, chunk_messages=per)
    for k, (pts, ring) in enumerate(clouds):
        t = 1000.0 + 0.1 * k
        w.write("/velodyne_points", "sensor_msgs/PointCloud2", t, rb.serialize_pointcloud2(t, "velodyne", pts, ring, k))
        if with_imu:
            w.write("/imu/data", "sensor_msgs/Imu", t + 0.01, rb.serialize_imu(t + 0.01, "imu", [0, 0, 0, 1], [0.1 * k, 0, 1], [0, k, 9.8]))
    w.close()


@pytest.mark.parametrize("comp", ["none", "bz2"])
def test_round_trip_topics_order_and_compression(tmp_path, comp):
    rng = np.random.default_rng(0)
    clouds = [(rng.normal(size=(50 + 7 * k, 4)).astype(np.float32), rng.integers(0, 16, 50 + 7 * k).astype(np.uint16)) for k in range(8)]
    p = str(tmp_path / "a.bag")
    _write(p, clouds, comp)
    got = list(rb.BagReader(p).messages())
    assert [g[0] for g in got] == ["/velodyne_points", "/imu/data"] * 8 and got[0][1] == "sensor_msgs/PointCloud2"
    assert all(abs(got[2 * k][2] - (1000.0 + 0.1 * k)) < 1e-6 for k in range(8))               # record times
    for k, (pts, ring) in enumerate(clouds):
        pc = rb.parse_pointcloud2(got[2 * k][3])
        a, b = rb.cloud_to_msfl(pc)
        assert np.array_equal(a, pts) and np.array_equal(b, ring) and pc["frame_id"] == "velodyne" and abs(pc["stamp"] - 1000.0 - 0.1 * k) < 1e-6
        imu = rb.parse_imu(got[2 * k + 1][3])
        assert imu["linear_acceleration"][1] == k and abs(imu["angular_velocity"][0] - 0.1 * k) < 1e-12
    only = list(rb.BagReader(p).messages(topics=["/imu/data"]))
    assert len(only) == 8 and all(t == "/imu/data" for t, _, _, _ in only)


def test_pointcloud2_layouts_and_refusals(tmp_path):
    # a hand-assembled message: two rows with row padding, fields in another order, an extra f64 `time` field
    n_w, n_h, step = 3, 2, 40
    hdr = struct.pack("<III", 5, 12, 500000000) + struct.pack("<I", 3) + b"vel"
    fields = [("intensity", 12, 7), ("z", 8, 7), ("y", 4, 7), ("x", 0, 7), ("ring", 16, 4), ("time", 24, 8)]
    body = hdr + struct.pack("<II", n_h, n_w) + struct.pack("<I", len(fields))
    for name, off, dt in fields:
        body += struct.pack("<I", len(name)) + name.encode() + struct.pack("<IBI", off, dt, 1)
    row_step = n_w * step + 16
    blob = bytearray(n_h * row_step)
    for r in range(n_h):
        for c in range(n_w):
            o = r * row_step + c * step
            blob[o:o + 16] = struct.pack("<4f", 1 + c, 10 + r, 100 + c + r, 0.5)
            blob[o + 16:o + 18] = struct.pack("<H", 3 * r + c)
            blob[o + 24:o + 32] = struct.pack("<d", 0.01 * c)
    body += struct.pack("<BII", 0, step, row_step) + struct.pack("<I", len(blob)) + bytes(blob) + b"\x01"
    pc = rb.parse_pointcloud2(body)
    pts, ring = rb.cloud_to_msfl(pc)
    assert pc["n"] == 6 and abs(pc["stamp"] - 12.5) < 1e-9 and list(ring) == [0, 1, 2, 3, 4, 5]
    assert np.array_equal(pts[4], np.array([2, 11, 102, 0.5], np.float32)) and pc["fields"]["time"][2] == 0.02
    no_ring = dict(pc, fields={k: v for k, v in pc["fields"].items() if k != "ring"})
    with pytest.raises(ValueError, match="ring"):
        rb.cloud_to_msfl(no_ring)
    # not a bag / unsupported chunk compression
    bad = tmp_path / "x.bag"
    bad.write_bytes(b"not a bag")
    with pytest.raises(ValueError):
        rb.BagReader(str(bad))
    p = str(tmp_path / "lz4.bag")
    _write(p, [(np.zeros((4, 4), np.float32), np.zeros(4, np.uint16))], per=1, with_imu=False)
    # patch the chunk record's compression field: length prefix 16 ("compression=none") -> 15 ("compression=lz4"), header one byte shorter
    raw = open(p, "rb").read()
    i = raw.index(b"compression=none")
    hl_pos = raw.rindex(struct.pack("<I", 16), 0, i)                 # the field's own length prefix sits right before it
    assert hl_pos == i - 4
    j = raw.rindex(b"\x04\x00\x00\x00op=\x05", 0, i)                 # field "op=\x05" (len 4) starts the chunk record's header
    (hl,) = struct.unpack_from("<I", raw, j - 4)
    patched = raw[:j - 4] + struct.pack("<I", hl - 1) + raw[j:i - 4] + struct.pack("<I", 15) + b"compression=lz4" + raw[i + 16:]
    open(p, "wb").write(patched)
    with pytest.raises(ValueError, match="lz4"):
        list(rb.BagReader(p).messages())


@pytest.mark.gpu
def test_bag_replay_equals_the_direct_replay(tmp_path):
    """examples/replay_bag.py on a bag of synthetic scans: the poses in its pose log are those of feeding the same scans to the SLAM
    step directly, bit for bit (the bag carries f32 points and u16 rings losslessly)."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import replay_synthetic as rp
    from msf_loam_amd import dataset
    world = synth.World(ground_half=45.0)
    truth = rp.trajectory(120)[:20]
    scans = [synth.make_scan(world, truth[k], synth.SEED + 5000 + k) for k in range(len(truth))]
    bag = str(tmp_path / "drive.bag")
    _write(bag, scans, comp="bz2", per=4)
    out = str(tmp_path / "poses.pb")
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "replay_bag.py"), bag, "--out", out, "--max-points", str(max(len(p) for p, _ in scans)),
                         "--rings", "16"], capture_output=True, text=True, timeout=300)
    assert cp.returncode == 0, cp.stderr[-2000:]
    log = dataset.PoseLog.load(out)
    from msf_loam_amd import capi
    slam = capi.Slam(0, max_scan_points=max(len(p) for p, _ in scans), max_rings=16)
    ref = np.array([np.array(slam.add_scan(*s).pose_map[:]) for s in scans])
    slam.close()
    got = np.array([p for _, p in log.odom])
    assert got.shape == ref.shape and np.array_equal(got, ref)
    assert len(log.imu) == 20
