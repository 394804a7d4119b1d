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
    # not a bag / a chunk labelled lz4 whose payload is no LZ4 frame
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


def test_lz4_block_and_frame_known_answers():
    """Hand-assembled bytes from the published LZ4 block / frame formats: literals + an overlapping match, 255-extended
    literal and match lengths, a run (offset 1), a stored block, the end mark; and the refusals."""
    # "abc" + match(offset 3, length 9: overlaps its own output) + last literals
    assert rb.lz4_block_decompress(bytes([0x35]) + b"abc" + bytes([3, 0]) + bytes([0x50]) + b"xyz12") == b"abc" + b"abc" * 3 + b"xyz12"
    # 273 literals: 15 in the token, then 255 + 3
    lit = bytes(range(256)) + bytes(range(17))
    assert rb.lz4_block_decompress(bytes([0xF0, 255, 3]) + lit) == lit
    # a run: one literal, match offset 1 of length 300 (15 in the token + 255 + 26, + 4), then five literals
    assert rb.lz4_block_decompress(bytes([0x1F]) + b"a" + bytes([1, 0, 255, 26]) + bytes([0x50]) + b"bbbbb") == b"a" * 301 + b"bbbbb"
    # match reaching back across an earlier match
    assert rb.lz4_block_decompress(bytes([0x40]) + b"0123" + bytes([4, 0]) + bytes([0x00, 8, 0]) + bytes([0x50]) + b"ABCDE") == b"0123" * 2 + b"0123" + b"ABCDE"
    for bad in (bytes([0x10]) + b"a" + bytes([2, 0]) + bytes([0x50]) + b"bbbbb",       # offset beyond the output
                bytes([0x10]) + b"a" + bytes([0, 0]) + bytes([0x50]) + b"bbbbb",       # offset 0
                bytes([0xF0, 255]),                                                    # truncated length
                bytes([0x30]) + b"ab"):                                                # literals past the end
        with pytest.raises(ValueError, match="lz4 block"):
            rb.lz4_block_decompress(bad)
    # a frame by hand: magic, FLG = version 1 | independent blocks, BD = 64 KB, header checksum, one compressed and one stored block, end mark
    try:
        import xxhash
        hc = (xxhash.xxh32(bytes([0x60, 0x40]), seed=0).intdigest() >> 8) & 0xff
    except ImportError:
        hc = 0
    blk = bytes([0x35]) + b"abc" + bytes([3, 0]) + bytes([0x50]) + b"xyz12"
    frame = struct.pack("<I", 0x184D2204) + bytes([0x60, 0x40, hc]) + struct.pack("<I", len(blk)) + blk + struct.pack("<I", 5 | 0x80000000) + b"STORE" + struct.pack("<I", 0)
    assert rb.lz4_frame_decompress(frame) == b"abc" * 4 + b"xyz12" + b"STORE"
    with pytest.raises(ValueError, match="record says"):
        rb.lz4_frame_decompress(frame, expect_size=3)
    with pytest.raises(ValueError, match="magic"):
        rb.lz4_frame_decompress(b"\x00" * 16)
    with pytest.raises(ValueError, match="truncated"):
        rb.lz4_frame_decompress(frame[:-4])
    # the module's own encoder (used by the writer for fixtures) against the decoder on awkward inputs
    rng = np.random.default_rng(1)
    for data in (b"", b"x", b"hello hello hello hello", bytes(rng.integers(0, 3, 20000, dtype=np.uint8)), bytes(rng.integers(0, 256, 70000, dtype=np.uint8)), b"abcdefgh" * 20000):
        assert rb.lz4_frame_decompress(rb.lz4_frame_compress(data), len(data)) == data


def test_lz4_block_linked_frames_and_block_checksums():
    """ADVICE r05.  liblz4's LZ4F default (and the lz4 command line tool) writes block-LINKED frames: FLG bit 5 clear, a match may reach
    into the up to 64 KB of output before its block.  Hand-assembled known answer, the refusal when the same blocks are labelled
    independent, per-block checksums verified, and the module's encoder / decoder pair on inputs whose repetitions span blocks."""
    try:
        import xxhash
    except ImportError:
        xxhash = None
    def header(flg, bd=0x40):
        hc = (xxhash.xxh32(bytes([flg, bd]), seed=0).intdigest() >> 8) & 0xff if xxhash else 0
        return struct.pack("<I", 0x184D2204) + bytes([flg, bd, hc])
    b1 = bytes([0x80]) + b"abcdefgh"                                             # eight literals, end of block
    b2 = bytes([0x04, 8, 0]) + bytes([0x50]) + b"xyz12"                          # no literals + match(offset 8, length 8) INTO block 1, then the last literals
    body = struct.pack("<I", len(b1)) + b1 + struct.pack("<I", len(b2)) + b2 + struct.pack("<I", 0)
    assert rb.lz4_frame_decompress(header(0x40) + body) == b"abcdefgh" * 2 + b"xyz12"
    with pytest.raises(ValueError, match="history"):
        rb.lz4_frame_decompress(header(0x60) + body)                             # the same blocks labelled independent: no history to reach into
    # a match that starts in the history and runs on into its own output (offset 8, length 20 over an 8-byte history)
    b3 = bytes([0x0F, 8, 0, 1]) + bytes([0x50]) + b"xyz12"
    assert rb.lz4_frame_decompress(header(0x40) + struct.pack("<I", len(b1)) + b1 + struct.pack("<I", len(b3)) + b3 + struct.pack("<I", 0)) == b"abcdefgh" + (b"abcdefgh" * 3)[:20] + b"xyz12"
    with pytest.raises(ValueError, match="dictionary"):
        rb.lz4_frame_decompress(header(0x41) + struct.pack("<I", 0) + body)
    if xxhash:
        def ck(b): return struct.pack("<I", xxhash.xxh32(b, seed=0).intdigest())
        good = header(0x50) + struct.pack("<I", len(b1)) + b1 + ck(b1) + struct.pack("<I", len(b2)) + b2 + ck(b2) + struct.pack("<I", 0)
        assert rb.lz4_frame_decompress(good) == b"abcdefgh" * 2 + b"xyz12"
        bad = bytearray(good); bad[7 + 4 + len(b1)] ^= 1
        with pytest.raises(ValueError, match="block checksum"):
            rb.lz4_frame_decompress(bytes(bad))
    rng = np.random.default_rng(2)
    for data in (b"abcdefgh" * 20000, bytes(rng.integers(0, 4, 200000, dtype=np.uint8)), bytes(rng.integers(0, 256, 70000, dtype=np.uint8)) * 2, b"q" * 65537):
        for kw in (dict(linked=True), dict(linked=True, block_checksums=True), dict(block_checksums=True), dict(linked=True, block=4 << 10)):
            f = rb.lz4_frame_compress(data, **kw)
            assert rb.lz4_frame_decompress(f, len(data)) == data
    # the linked encoder really uses the history (otherwise the fixture proves nothing): a block that repeats the previous one shrinks to a few bytes
    rep = bytes(rng.integers(0, 256, 4096, dtype=np.uint8)) * 4
    assert len(rb.lz4_frame_compress(rep, block=4096, linked=True)) < 4096 + 600 < len(rb.lz4_frame_compress(rep, block=4096))


def test_lz4_bag_round_trip_and_time_merge(tmp_path):
    rng = np.random.default_rng(3)
    clouds = [(rng.normal(size=(300 + 11 * k, 4)).astype(np.float32), rng.integers(0, 16, 300 + 11 * k).astype(np.uint16)) for k in range(7)]
    p = str(tmp_path / "l.bag")
    _write(p, clouds, "lz4", per=3)
    assert b"compression=lz4" in open(p, "rb").read()
    got = list(rb.BagReader(p).messages())
    assert len(got) == 14
    for k, (pts, ring) in enumerate(clouds):
        a, b = rb.cloud_to_msfl(rb.parse_pointcloud2(got[2 * k][3]))
        assert np.array_equal(a, pts) and np.array_equal(b, ring)
    # a bag whose chunks are NOT in time order (re-indexed / merged bags): IMU written first, then the clouds
    q = str(tmp_path / "m.bag")
    w = rb.BagWriter(q, compression="lz4", chunk_messages=4)
    for k in range(6):
        w.write("/imu/data", "sensor_msgs/Imu", 1000.05 + 0.1 * k, rb.serialize_imu(1000.05 + 0.1 * k, "imu", [0, 0, 0, 1], [k, 0, 0], [0, 0, 9.8]))
    for k, (pts, ring) in enumerate(clouds[:6]):
        w.write("/velodyne_points", "sensor_msgs/PointCloud2", 1000.0 + 0.1 * k, rb.serialize_pointcloud2(1000.0 + 0.1 * k, "velodyne", pts, ring, k))
    w.close()
    in_file = [g[0] for g in rb.BagReader(q).messages(by_time=False)]
    assert in_file == ["/imu/data"] * 6 + ["/velodyne_points"] * 6
    merged = list(rb.BagReader(q).messages())                         # like rosbag::View: by record time
    assert [g[0] for g in merged] == ["/velodyne_points", "/imu/data"] * 6
    assert all(merged[i][2] <= merged[i + 1][2] for i in range(11))
    assert rb.parse_imu(merged[3][3])["angular_velocity"][0] == 1.0
    only = list(rb.BagReader(q).messages(topics=["/velodyne_points"]))
    assert len(only) == 6 and np.array_equal(rb.cloud_to_msfl(rb.parse_pointcloud2(only[5][3]))[0], clouds[5][0])


def test_truncated_files_and_messages_raise_value_errors(tmp_path):
    rng = np.random.default_rng(4)
    clouds = [(rng.normal(size=(100, 4)).astype(np.float32), rng.integers(0, 16, 100).astype(np.uint16)) for _ in range(4)]
    p = str(tmp_path / "t.bag")
    _write(p, clouds, "none", per=2)
    raw = open(p, "rb").read()
    for cut in (len(raw) - 7, len(raw) - 3000, 4096 + 9):
        q = tmp_path / ("cut%d.bag" % cut)
        q.write_bytes(raw[:cut])
        with pytest.raises(ValueError):
            list(rb.BagReader(str(q)).messages())
    msg = rb.serialize_pointcloud2(1.0, "v", *clouds[0])
    with pytest.raises(ValueError, match="truncated"):
        rb.parse_pointcloud2(msg[:-900])                              # the data blob is shorter than width x point_step


def test_velodyne_driver_point_layouts():
    """The two layouts Velodyne's ROS driver has published, assembled by hand:
    packed PointXYZIRT  x y z intensity @0/4/8/12, ring u16 @16, time f32 @18 (UNALIGNED), point_step 22;
    PCL-aligned         x y z @0/4/8, intensity @16, ring u16 @20, time f32 @24, point_step 32 (the reference's struct, common.h:44-62).
    Both must give the same (x y z intensity, ring) arrays; the reference reads by field NAME through pcl::fromROSMsg."""
    rng = np.random.default_rng(5)
    n = 37
    xyz_i = rng.normal(size=(n, 4)).astype(np.float32)
    ring = rng.integers(0, 16, n).astype(np.uint16)
    tm = rng.uniform(0, 0.1, n).astype(np.float32)

    def msg(fields, step, fill):
        hdr = struct.pack("<III", 1, 100, 0) + struct.pack("<I", 8) + b"velodyne"
        body = hdr + struct.pack("<II", 1, n) + struct.pack("<I", len(fields))
        for name, off, dt in fields:
            body += struct.pack("<I", len(name)) + name.encode() + struct.pack("<IBI", off, dt, 1)
        blob = bytearray(n * step)
        for k in range(n):
            fill(blob, k * step, k)
        return body + struct.pack("<BII", 0, step, step * n) + struct.pack("<I", len(blob)) + bytes(blob) + b"\x01"

    def fill22(b, o, k):
        b[o:o + 16] = xyz_i[k].tobytes(); b[o + 16:o + 18] = struct.pack("<H", ring[k]); b[o + 18:o + 22] = struct.pack("<f", tm[k])

    def fill32(b, o, k):
        b[o:o + 12] = xyz_i[k, :3].tobytes(); b[o + 16:o + 20] = xyz_i[k, 3:].tobytes(); b[o + 20:o + 22] = struct.pack("<H", ring[k]); b[o + 24:o + 28] = struct.pack("<f", tm[k])
    packed = rb.parse_pointcloud2(msg([("x", 0, 7), ("y", 4, 7), ("z", 8, 7), ("intensity", 12, 7), ("ring", 16, 4), ("time", 18, 7)], 22, fill22))
    aligned = rb.parse_pointcloud2(msg([("x", 0, 7), ("y", 4, 7), ("z", 8, 7), ("intensity", 16, 7), ("ring", 20, 4), ("time", 24, 7)], 32, fill32))
    for pc in (packed, aligned):
        pts, r = rb.cloud_to_msfl(pc)
        assert np.array_equal(pts, xyz_i) and np.array_equal(r, ring) and np.array_equal(pc["fields"]["time"], tm)
    with pytest.raises(ValueError, match="does not fit"):
        rb.parse_pointcloud2(msg([("x", 0, 7), ("y", 4, 7), ("z", 8, 7), ("ring", 21, 4)], 22, fill22))


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
