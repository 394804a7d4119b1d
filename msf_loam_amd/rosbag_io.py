"""ROS-free reader (and a minimal writer, for tests and hand-made fixtures) of rosbag format 2.0 — enough of it to replay
BASELINE configs[2] (`nsh_indoor_outdoor.bag`, reference README.md:42-47) through the C ABI when the bag is available:
`sensor_msgs/PointCloud2` scans (the reference's `/velodyne_points`, msf_loam_node.cc:442-460 via rosbag::View) and
`sensor_msgs/Imu` samples (`/imu/data`), in recorded order.  SURVEY.md 8f N4.  Host-side I/O only; the compute is libmsfl_hip.so's.

Bag format 2.0 ([3P-recall] of the published ROS wiki page "Bags/Format/2.0"; no ROS package is imported):
    "#ROSBAG V2.0\\n", then records  <header_len u32><header><data_len u32><data>
    header = fields  <len u32><name>=<value>;  op (1 byte): 0x03 bag header, 0x05 chunk, 0x07 connection, 0x02 message data,
             0x04 index data, 0x06 chunk info
    chunk  : compression = "none" | "bz2" | "lz4", size = uncompressed bytes; data = the concatenated connection / message records
    message: conn u32, time (secs u32, nsecs u32); data = the serialised ROS message (little endian)
lz4 chunks are refused with a clear error (no lz4 decoder in the Python standard library); `rosbag decompress` on the maintainer's side.
"""
import bz2
import struct

import numpy as np

_MAGIC = b"#ROSBAG V2.0\n"
OP_MSG, OP_BAG_HEADER, OP_INDEX, OP_CHUNK, OP_CHUNK_INFO, OP_CONNECTION = 0x02, 0x03, 0x04, 0x05, 0x06, 0x07
# sensor_msgs/PointField datatypes -> numpy
_PF = {1: "i1", 2: "u1", 3: "<i2", 4: "<u2", 5: "<i4", 6: "<u4", 7: "<f4", 8: "<f8"}


def _parse_header(buf):
    out, i = {}, 0
    while i < len(buf):
        (n,) = struct.unpack_from("<I", buf, i)
        field = buf[i + 4:i + 4 + n]
        i += 4 + n
        k, _, v = field.partition(b"=")
        out[k.decode()] = v
    return out


def _records(buf):
    """(header dict, data bytes) of the records packed in `buf`."""
    i = 0
    while i + 4 <= len(buf):
        (hl,) = struct.unpack_from("<I", buf, i)
        header = _parse_header(buf[i + 4:i + 4 + hl])
        i += 4 + hl
        (dl,) = struct.unpack_from("<I", buf, i)
        yield header, buf[i + 4:i + 4 + dl]
        i += 4 + dl


class _Cursor:
    def __init__(self, b):
        self.b, self.i = b, 0

    def take(self, fmt):
        v = struct.unpack_from("<" + fmt, self.b, self.i)
        self.i += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def string(self):
        n = self.take("I")
        s = self.b[self.i:self.i + n]
        self.i += n
        return s

    def header(self):
        """std_msgs/Header -> (seq, stamp seconds as float, frame_id)"""
        seq, secs, nsecs = self.take("I"), self.take("I"), self.take("I")
        return seq, secs + 1e-9 * nsecs, self.string().decode(errors="replace")


def parse_pointcloud2(data):
    """Serialised sensor_msgs/PointCloud2 -> dict(stamp, frame_id, fields {name: array (n,)}, n)."""
    c = _Cursor(data)
    _, stamp, frame = c.header()
    height, width = c.take("I"), c.take("I")
    fields = []
    for _ in range(c.take("I")):
        name = c.string().decode()
        offset, datatype, count = c.take("I"), c.take("B"), c.take("I")
        fields.append((name, offset, datatype, count))
    big, point_step, row_step = c.take("B"), c.take("I"), c.take("I")
    blob = c.string()
    if big:
        raise ValueError("big-endian PointCloud2 is not supported")
    n = height * width
    raw = np.frombuffer(blob, dtype=np.uint8)
    if row_step != width * point_step:                      # padded rows
        raw = raw.reshape(height, row_step)[:, :width * point_step].reshape(-1)
    raw = raw[:n * point_step].reshape(n, point_step)
    out = {}
    for name, offset, datatype, count in fields:
        dt = np.dtype(_PF[datatype])
        col = np.ascontiguousarray(raw[:, offset:offset + dt.itemsize]).view(dt).reshape(n)
        out[name] = col
    return dict(stamp=stamp, frame_id=frame, fields=out, n=n)


def cloud_to_msfl(pc, ring_field="ring"):
    """PointCloud2 dict -> (pts (n,4) float32 [x y z intensity], ring (n,) uint16): the inputs of msfl_extract_features /
    msfl_slam_add_scan, in driver order like pcl::fromROSMsg delivers them (msf_loam_node.cc:166-167).  The extraction needs the
    driver's ring field (CHECK at msf_loam_node.cc:136); a bag without one is an error here, not a guess."""
    f = pc["fields"]
    for k in ("x", "y", "z"):
        if k not in f:
            raise ValueError("PointCloud2 without field %r" % k)
    if ring_field not in f:
        raise ValueError("PointCloud2 without a %r field: the feature extraction needs the driver's ring ids" % ring_field)
    pts = np.zeros((pc["n"], 4), np.float32)
    pts[:, 0], pts[:, 1], pts[:, 2] = f["x"], f["y"], f["z"]
    if "intensity" in f:
        pts[:, 3] = f["intensity"].astype(np.float32)
    return pts, f[ring_field].astype(np.uint16)


def parse_imu(data):
    """Serialised sensor_msgs/Imu -> dict(stamp, orientation xyzw, angular_velocity, linear_acceleration)."""
    c = _Cursor(data)
    _, stamp, frame = c.header()
    q = np.array(c.take("4d")); c.take("9d")
    w = np.array(c.take("3d")); c.take("9d")
    a = np.array(c.take("3d"))
    return dict(stamp=stamp, frame_id=frame, orientation=q, angular_velocity=w, linear_acceleration=a)


class BagReader:
    """Sequential reader: `for topic, msgtype, t, data in BagReader(path).messages(topics=[...])`."""

    def __init__(self, path):
        self.path = path
        with open(path, "rb") as f:
            if f.read(len(_MAGIC)) != _MAGIC:
                raise ValueError("%s is not a rosbag 2.0 file" % path)

    def messages(self, topics=None):
        conns = {}
        want = set(topics) if topics else None
        with open(self.path, "rb") as f:
            f.seek(len(_MAGIC))
            while True:
                head = f.read(4)
                if len(head) < 4:
                    return
                (hl,) = struct.unpack("<I", head)
                header = _parse_header(f.read(hl))
                (dl,) = struct.unpack("<I", f.read(4))
                op = header["op"][0]
                if op == OP_CHUNK:
                    comp = header["compression"].decode()
                    blob = f.read(dl)
                    if comp == "bz2":
                        blob = bz2.decompress(blob)
                    elif comp != "none":
                        raise ValueError("chunk compression %r is not supported (only none / bz2): run `rosbag decompress` first" % comp)
                    for h2, d2 in _records(blob):
                        yield from self._record(h2, d2, conns, want)
                elif op in (OP_CONNECTION, OP_MSG):
                    yield from self._record(header, f.read(dl), conns, want)
                else:                                   # bag header, index data, chunk info: not needed for a sequential read
                    f.seek(dl, 1)

    @staticmethod
    def _record(header, data, conns, want):
        op = header["op"][0]
        if op == OP_CONNECTION:
            (cid,) = struct.unpack("<I", header["conn"])
            info = _parse_header(data)
            conns[cid] = (header["topic"].decode(), info.get("type", b"").decode())
        elif op == OP_MSG:
            (cid,) = struct.unpack("<I", header["conn"])
            secs, nsecs = struct.unpack("<II", header["time"])
            topic, mtype = conns.get(cid, ("?", "?"))
            if want is None or topic in want:
                yield topic, mtype, secs + 1e-9 * nsecs, data
        return


# ---------------------------------------------------------------------------------------------- minimal writer (tests, fixtures)

def _field(name, value):
    b = name.encode() + b"=" + value
    return struct.pack("<I", len(b)) + b


def _record(header_fields, data):
    h = b"".join(_field(k, v) for k, v in header_fields)
    return struct.pack("<I", len(h)) + h + struct.pack("<I", len(data)) + data


def serialize_pointcloud2(stamp, frame_id, pts, ring, seq=0):
    """The layout velodyne_pointcloud publishes for PointXYZIR: x, y, z @0/4/8, intensity @16, ring (u16) @20, point_step 32."""
    n = len(pts)
    secs = int(stamp); nsecs = int(round((stamp - secs) * 1e9))
    out = struct.pack("<III", seq, secs, nsecs) + struct.pack("<I", len(frame_id)) + frame_id.encode()
    out += struct.pack("<II", 1, n)
    fields = [("x", 0, 7), ("y", 4, 7), ("z", 8, 7), ("intensity", 16, 7), ("ring", 20, 4)]
    out += struct.pack("<I", len(fields))
    for name, off, dt in fields:
        out += struct.pack("<I", len(name)) + name.encode() + struct.pack("<IBI", off, dt, 1)
    step = 32
    blob = np.zeros((n, step), np.uint8)
    p = np.ascontiguousarray(pts, np.float32)
    blob[:, 0:12] = p[:, :3].copy().view(np.uint8).reshape(n, 12)
    blob[:, 16:20] = np.ascontiguousarray(p[:, 3]).view(np.uint8).reshape(n, 4)
    blob[:, 20:22] = np.ascontiguousarray(ring, np.uint16).view(np.uint8).reshape(n, 2)
    out += struct.pack("<BII", 0, step, step * n) + struct.pack("<I", blob.size) + blob.tobytes() + struct.pack("<B", 1)
    return out


def serialize_imu(stamp, frame_id, orientation, angular_velocity, linear_acceleration, seq=0):
    secs = int(stamp); nsecs = int(round((stamp - secs) * 1e9))
    out = struct.pack("<III", seq, secs, nsecs) + struct.pack("<I", len(frame_id)) + frame_id.encode()
    z9 = struct.pack("<9d", *([0.0] * 9))
    return out + struct.pack("<4d", *orientation) + z9 + struct.pack("<3d", *angular_velocity) + z9 + struct.pack("<3d", *linear_acceleration) + z9


class BagWriter:
    """Messages -> a valid sequentially readable bag (one chunk per `chunk_messages` messages; no index records)."""

    def __init__(self, path, compression="none", chunk_messages=8):
        self.f = open(path, "wb")
        self.comp, self.per = compression, chunk_messages
        self.conns, self.buf, self.count = {}, b"", 0
        self.f.write(_MAGIC)
        head = _record([("op", bytes([OP_BAG_HEADER])), ("index_pos", struct.pack("<Q", 0)), ("conn_count", struct.pack("<I", 0)),
                        ("chunk_count", struct.pack("<I", 0))], b"")
        self.f.write(head[:-4] + struct.pack("<I", 4096 - len(head)) + b" " * (4096 - len(head)))   # padded to 4096 like rosbag does

    def write(self, topic, msgtype, t, data):
        if topic not in self.conns:
            cid = len(self.conns)
            self.conns[topic] = cid
            info = _field("topic", topic.encode()) + _field("type", msgtype.encode()) + _field("md5sum", b"0" * 32) + _field("message_definition", b"")
            self.buf += _record([("op", bytes([OP_CONNECTION])), ("conn", struct.pack("<I", cid)), ("topic", topic.encode())], info)
        secs = int(t); nsecs = int(round((t - secs) * 1e9))
        self.buf += _record([("op", bytes([OP_MSG])), ("conn", struct.pack("<I", self.conns[topic])), ("time", struct.pack("<II", secs, nsecs))], data)
        self.count += 1
        if self.count % self.per == 0:
            self._flush()

    def _flush(self):
        if not self.buf:
            return
        blob = bz2.compress(self.buf) if self.comp == "bz2" else self.buf
        self.f.write(_record([("op", bytes([OP_CHUNK])), ("compression", self.comp.encode()), ("size", struct.pack("<I", len(self.buf)))], blob))
        self.buf = b""

    def close(self):
        self._flush()
        self.f.close()
