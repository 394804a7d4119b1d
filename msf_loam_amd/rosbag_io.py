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
lz4 chunks: roslz4 writes the LZ4 FRAME format (magic 0x184D2204, FLG / BD / header-checksum bytes, blocks of <size u32><data> with
the top size bit marking a stored block, a zero size as end mark, optionally an xxh32 content checksum) around LZ4 BLOCK data (token:
literal length | match length - 4, 255-extended lengths, literals, u16 offset, overlapping match copy) — [3P-recall] of the published
lz4 "Frame format" and "Block format" documents; `lz4_frame_decompress` below is a dependency-free decoder of both (round 5).

Replay order: `BagReader.messages(..., by_time=True)` (the default) yields the messages of all requested connections merged by their
record time (ties in file order), which is what the reference's `rosbag::View` iteration does (msf_loam_node.cc:445-448); a bag written
strictly in time order reads the same either way, a re-indexed / filtered / merged one does not.  Truncated files raise ValueError.
"""
import bz2
import struct

import numpy as np

_MAGIC = b"#ROSBAG V2.0\n"
_LZ4_MAGIC = 0x184D2204


def lz4_block_decompress(src, max_out=None, prefix=b""):
    """One LZ4 block -> bytes.  Sequences of (token, [literal length extension], literals, offset u16, [match length extension]); the
    last sequence ends after its literals.  A match may overlap its own output (offset < length: a run).  `prefix`: the output that
    precedes this block in a block-LINKED frame (its last 64 KB are all a match can reach); the returned bytes are the block's own."""
    out = bytearray(prefix)
    base = len(out)
    i, n = 0, len(src)
    while i < n:
        token = src[i]; i += 1
        lit = token >> 4
        if lit == 15:
            while True:
                if i >= n:
                    raise ValueError("lz4 block: truncated literal length")
                b = src[i]; i += 1
                lit += b
                if b != 255:
                    break
        if i + lit > n:
            raise ValueError("lz4 block: literals run past the end of the block")
        out += src[i:i + lit]; i += lit
        if i >= n:
            break                                            # the last sequence has no match part
        if i + 2 > n:
            raise ValueError("lz4 block: truncated match offset")
        off = src[i] | (src[i + 1] << 8); i += 2
        if off == 0 or off > len(out):
            raise ValueError("lz4 block: match offset %d outside the %d bytes decoded so far (%d of them history of earlier blocks)" % (off, len(out), base))
        ml = token & 15
        if ml == 15:
            while True:
                if i >= n:
                    raise ValueError("lz4 block: truncated match length")
                b = src[i]; i += 1
                ml += b
                if b != 255:
                    break
        ml += 4
        start = len(out) - off
        if off >= ml:
            out += out[start:start + ml]
        else:                                                # overlapping copy: the pattern of `off` bytes repeats
            pat = bytes(out[start:])
            out += (pat * (ml // off + 1))[:ml]
        if max_out is not None and len(out) - base > max_out:
            raise ValueError("lz4 block: output exceeds the declared size")
    return bytes(out[base:])


def lz4_frame_decompress(buf, expect_size=None):
    """One LZ4 frame (what a rosbag `compression=lz4` chunk holds) -> bytes."""
    if len(buf) < 7 or struct.unpack_from("<I", buf, 0)[0] != _LZ4_MAGIC:
        raise ValueError("lz4 frame: bad magic number")
    flg, bd = buf[4], buf[5]
    if (flg >> 6) != 1:
        raise ValueError("lz4 frame: unsupported version %d" % (flg >> 6))
    independent, block_checksum, has_size, content_checksum, has_dict = (flg >> 5) & 1, (flg >> 4) & 1, (flg >> 3) & 1, (flg >> 2) & 1, flg & 1
    if has_dict:
        raise ValueError("lz4 frame: frames that need an external dictionary are not supported")
    i = 6
    content_size = None
    if has_size:
        (content_size,) = struct.unpack_from("<Q", buf, i); i += 8
    i += 1                                                    # header checksum byte ((xxh32(descriptor) >> 8) & 0xff): checked when xxhash is there
    try:
        import xxhash
        if buf[i - 1] != (xxhash.xxh32(bytes(buf[4:i - 1]), seed=0).intdigest() >> 8) & 0xff:
            raise ValueError("lz4 frame: header checksum mismatch")
    except ImportError:
        xxhash = None
    block_max = {4: 64 << 10, 5: 256 << 10, 6: 1 << 20, 7: 4 << 20}.get((bd >> 4) & 7)
    if block_max is None:
        raise ValueError("lz4 frame: bad block size id %d" % ((bd >> 4) & 7))
    # Block-LINKED frames (FLG bit 5 clear: liblz4's LZ4F default, what the lz4 command line tool writes) let a match reach into the
    # up to 64 KB of output before its block; roslz4 writes independent blocks.  Both are decoded.
    res = bytearray()
    while True:
        if i + 4 > len(buf):
            raise ValueError("lz4 frame: truncated (no end mark)")
        (sz,) = struct.unpack_from("<I", buf, i); i += 4
        if sz == 0:
            break
        stored, sz = sz >> 31, sz & 0x7fffffff
        if i + sz > len(buf):
            raise ValueError("lz4 frame: block runs past the end of the chunk")
        data = buf[i:i + sz]; i += sz
        if block_checksum:
            if i + 4 > len(buf):
                raise ValueError("lz4 frame: truncated block checksum")
            if xxhash is not None and struct.unpack_from("<I", buf, i)[0] != xxhash.xxh32(bytes(data), seed=0).intdigest():
                raise ValueError("lz4 frame: block checksum mismatch")
            i += 4
        if stored:
            if sz > block_max:
                raise ValueError("lz4 frame: stored block larger than the frame's block size")
            res += data
        else:
            res += lz4_block_decompress(data, block_max, b"" if independent else bytes(res[-65536:]))
    res = bytes(res)
    if content_checksum:
        if i + 4 > len(buf):
            raise ValueError("lz4 frame: truncated content checksum")
        if xxhash is not None and struct.unpack_from("<I", buf, i)[0] != xxhash.xxh32(res, seed=0).intdigest():
            raise ValueError("lz4 frame: content checksum mismatch")
    if content_size is not None and content_size != len(res):
        raise ValueError("lz4 frame: content size %d != %d decoded" % (content_size, len(res)))
    if expect_size is not None and expect_size != len(res):
        raise ValueError("lz4 chunk: %d bytes decoded, the chunk record says %d" % (len(res), expect_size))
    return res


def _lz4_block_compress(data, prefix=b""):
    """Greedy LZ4 block encoder for the writer / fixtures (4-byte hash table, first match wins): valid, not tuned.  `prefix`: history a
    block of a LINKED frame may refer to (the <= 64 KB of input before it)."""
    base = len(prefix)
    data = bytes(prefix) + bytes(data)
    n = len(data)
    out = bytearray()
    table = {}
    for j in range(0, max(base - 3, 0)):
        table[data[j:j + 4]] = j
    anchor = i = base
    def emit(lit_end, ml=None, off=0):
        lit = lit_end - anchor
        token = (min(lit, 15) << 4) | (0 if ml is None else min(ml - 4, 15))
        out.append(token)
        if lit >= 15:
            r = lit - 15
            while r >= 255:
                out.append(255); r -= 255
            out.append(r)
        out.extend(data[anchor:lit_end])
        if ml is not None:
            out.extend(struct.pack("<H", off))
            if ml - 4 >= 15:
                r = ml - 4 - 15
                while r >= 255:
                    out.append(255); r -= 255
                out.append(r)
    while i + 4 <= n - 5:                                     # the last 5 bytes are always literals (format rule)
        key = data[i:i + 4]
        j = table.get(key)
        table[key] = i
        if j is not None and i - j <= 65535:
            ml = 4
            while i + ml < n - 5 and data[j + ml] == data[i + ml]:
                ml += 1
            emit(i, ml, i - j)
            i += ml
            anchor = i
        else:
            i += 1
    emit(n)
    return bytes(out)


def lz4_frame_compress(data, block=64 << 10, linked=False, block_checksums=False):
    """An LZ4 frame like roslz4 writes (version 1, independent blocks, content checksum when xxhash is importable).  `linked`: blocks
    that refer to the 64 KB before them (liblz4's LZ4F default; for the reader's fixtures); `block_checksums`: xxh32 after each block."""
    try:
        import xxhash
    except ImportError:
        xxhash = None
    block_checksums = bool(block_checksums and xxhash)
    flg = (1 << 6) | (0 if linked else (1 << 5)) | ((1 << 4) if block_checksums else 0) | ((1 << 2) if xxhash else 0)
    desc = bytes([flg, 4 << 4])
    hc = (xxhash.xxh32(desc, seed=0).intdigest() >> 8) & 0xff if xxhash else 0
    out = [struct.pack("<I", _LZ4_MAGIC), desc, bytes([hc])]
    for o in range(0, len(data), block):
        raw = data[o:o + block]
        c = _lz4_block_compress(raw, data[max(o - 65536, 0):o] if linked else b"")
        if len(c) < len(raw):
            out += [struct.pack("<I", len(c)), c]
        else:
            c = raw
            out += [struct.pack("<I", len(raw) | 0x80000000), raw]
        if block_checksums:
            out.append(struct.pack("<I", xxhash.xxh32(bytes(c), seed=0).intdigest()))
    out.append(struct.pack("<I", 0))
    if xxhash:
        out.append(struct.pack("<I", xxhash.xxh32(data, seed=0).intdigest()))
    return b"".join(out)
OP_MSG, OP_BAG_HEADER, OP_INDEX, OP_CHUNK, OP_CHUNK_INFO, OP_CONNECTION = 0x02, 0x03, 0x04, 0x05, 0x06, 0x07
# sensor_msgs/PointField datatypes -> numpy
_PF = {1: "i1", 2: "u1", 3: "<i2", 4: "<u2", 5: "<i4", 6: "<u4", 7: "<f4", 8: "<f8"}


def _parse_header(buf):
    out, i = {}, 0
    while i < len(buf):
        if i + 4 > len(buf):
            raise ValueError("rosbag record header: truncated field length")
        (n,) = struct.unpack_from("<I", buf, i)
        if i + 4 + n > len(buf):
            raise ValueError("rosbag record header: field runs past the header")
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
        if i + 4 + hl + 4 > len(buf):
            raise ValueError("rosbag chunk: truncated record header")
        header = _parse_header(buf[i + 4:i + 4 + hl])
        i += 4 + hl
        (dl,) = struct.unpack_from("<I", buf, i)
        if i + 4 + dl > len(buf):
            raise ValueError("rosbag chunk: record data runs past the chunk")
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
    if row_step < width * point_step or len(raw) < (height * row_step if row_step != width * point_step else n * point_step):
        raise ValueError("PointCloud2: %d bytes of data for %d x %d points of %d bytes (row_step %d): truncated message"
                         % (len(raw), height, width, point_step, row_step))
    for name, offset, datatype, count in fields:
        if datatype not in _PF or offset + np.dtype(_PF[datatype]).itemsize > point_step:
            raise ValueError("PointCloud2: field %r (datatype %d at offset %d) does not fit a %d-byte point" % (name, datatype, offset, point_step))
    if row_step != width * point_step:                      # padded rows
        raw = raw[:height * row_step].reshape(height, row_step)[:, :width * point_step].reshape(-1)
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

    def _top_level(self, f):
        """(op, header, data length, data file position) of the file's top-level records; the caller reads or skips the data."""
        f.seek(0, 2)
        size = f.tell()
        f.seek(len(_MAGIC))
        while True:
            head = f.read(4)
            if len(head) == 0:
                return
            if len(head) < 4:
                raise ValueError("%s: truncated record (header length)" % self.path)
            (hl,) = struct.unpack("<I", head)
            hb = f.read(hl)
            dlb = f.read(4)
            if len(hb) < hl or len(dlb) < 4:
                raise ValueError("%s: truncated record header" % self.path)
            header = _parse_header(hb)
            (dl,) = struct.unpack("<I", dlb)
            if "op" not in header:
                raise ValueError("%s: record without an op field" % self.path)
            pos = f.tell()
            if pos + dl > size:
                raise ValueError("%s: truncated record (%d data bytes announced, %d left in the file)" % (self.path, dl, size - pos))
            yield header["op"][0], header, dl, pos
            f.seek(pos + dl)

    def _read(self, f, pos, dl):
        f.seek(pos)
        blob = f.read(dl)
        if len(blob) < dl:
            raise ValueError("%s: truncated record (%d of %d data bytes)" % (self.path, len(blob), dl))
        return blob

    def _chunk(self, f, header, pos, dl):
        comp = header["compression"].decode()
        blob = self._read(f, pos, dl)
        size = struct.unpack("<I", header["size"])[0] if "size" in header else None
        if comp == "bz2":
            blob = bz2.decompress(blob)
        elif comp == "lz4":
            blob = lz4_frame_decompress(blob, size)
        elif comp != "none":
            raise ValueError("chunk compression %r is not supported (none / bz2 / lz4)" % comp)
        if size is not None and len(blob) != size:
            raise ValueError("%s: chunk of %d bytes, its record says %d" % (self.path, len(blob), size))
        return blob

    def messages(self, topics=None, by_time=True):
        """by_time (default): merged by record time like rosbag::View (ties in file order); False: file order, one pass."""
        want = set(topics) if topics else None
        conns = {}
        with open(self.path, "rb") as f:
            if not by_time:
                for op, header, dl, pos in self._top_level(f):
                    if op == OP_CHUNK:
                        for h2, d2 in _records(self._chunk(f, header, pos, dl)):
                            yield from self._record(h2, d2, conns, want)
                    elif op in (OP_CONNECTION, OP_MSG):
                        yield from self._record(header, self._read(f, pos, dl), conns, want)
                return
            # pass 1: (time, ordinal, chunk position or -1, record index inside it) of every wanted message
            index, ordinal = [], 0
            tops = []
            for op, header, dl, pos in self._top_level(f):
                if op == OP_CHUNK:
                    tops.append((header, pos, dl))
                    for k, (h2, d2) in enumerate(_records(self._chunk(f, header, pos, dl))):
                        for _ in self._record(h2, d2, conns, want):            # the connection table fills as a side effect
                            secs, nsecs = struct.unpack("<II", h2["time"])
                            index.append((secs, nsecs, ordinal, len(tops) - 1, k)); ordinal += 1
                elif op in (OP_CONNECTION, OP_MSG):
                    tops.append((header, pos, dl))
                    for _ in self._record(header, self._read(f, pos, dl) if op == OP_CONNECTION else b"", conns, want):
                        secs, nsecs = struct.unpack("<II", header["time"])
                        index.append((secs, nsecs, ordinal, len(tops) - 1, -1)); ordinal += 1
            index.sort()
            # pass 2: in time order; the records of the last few chunks stay decoded (a recorded bag is nearly ordered already)
            cache, cache_order = {}, []
            for secs, nsecs, _, ti, k in index:
                header, pos, dl = tops[ti]
                if k < 0:
                    yield from self._record(header, self._read(f, pos, dl), conns, want)
                    continue
                if ti not in cache:
                    cache[ti] = list(_records(self._chunk(f, header, pos, dl)))
                    cache_order.append(ti)
                    if len(cache_order) > 4:
                        del cache[cache_order.pop(0)]
                h2, d2 = cache[ti][k]
                yield from self._record(h2, d2, conns, want)

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
        blob = bz2.compress(self.buf) if self.comp == "bz2" else lz4_frame_compress(self.buf) if self.comp == "lz4" else self.buf
        self.f.write(_record([("op", bytes([OP_CHUNK])), ("compression", self.comp.encode()), ("size", struct.pack("<I", len(self.buf)))], blob))
        self.buf = b""

    def close(self):
        self._flush()
        self.f.close()
