"""N4: ROS-free dataset I/O and the pose log, so recorded data can be replayed
through the C ABI without rosbag / protoc.

  KITTI odometry layout   src/slam/kitti_helper.cc:21-31 (velodyne/*.bin, 16 B per point),
                          :64-93 (calib.txt line 5 "Tr: "), :97-119 (poses/NN.txt, times.txt; values go
                          through stof, i.e. single precision; Tl = Tr^-1 * Tc * Tr, rotation normalised)
  ring field              the reference leaves "todo write scan ring here" (kitti_helper.cc:152) although
                          extraction needs it (msf_loam_node.cc:136); `rings_from_elevation` bins the
                          elevation angle uniformly — ours, not the reference's
  pose log                proto/msg.proto:1-37 `PbData{imu_datas=1, odom_datas=2}` as written by
                          LaserMapping (laser_mapping.cc:117,251-254); proto3 wire format encoded by hand
  time                    UniversalTimeScaleClock, int64 nanosecond ticks (common/time.h:7-16, time.cc:5-17)
"""
import os
import struct

import numpy as np


# ------------------------------------------------------------------------------------ KITTI

def read_kitti_bin(path):
    """velodyne/NNNNNN.bin -> (n, 4) float32 [x y z intensity] (kitti_helper.cc:21-31, 146-154)."""
    raw = np.fromfile(path, dtype=np.float32)
    return raw[: (len(raw) // 4) * 4].reshape(-1, 4).copy()


def write_kitti_bin(path, pts):
    np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 4).tofile(path)


def _f32_row(line):
    """`stof` per token, stored in a double matrix (kitti_helper.cc:82-88, 105-111)."""
    return np.array([np.float32(tok) for tok in line.split()], dtype=np.float32).astype(np.float64)


def _quat_from_matrix(R):
    """Eigen's Quaternion(Matrix3) (Shepperd's branches), [x y z w]."""
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        s = np.sqrt(t + 1.0)
        w = 0.5 * s
        s = 0.5 / s
        return np.array([(R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s, w])
    i = 0
    if R[1, 1] > R[0, 0]:
        i = 1
    if R[2, 2] > R[i, i]:
        i = 2
    j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    q = np.zeros(4)
    q[i] = 0.5 * s
    s = 0.5 / s
    q[3] = (R[k, j] - R[j, k]) * s
    q[j] = (R[j, i] + R[i, j]) * s
    q[k] = (R[k, i] + R[i, k]) * s
    return q


def _pose7(R, t):
    q = _quat_from_matrix(R)
    return np.concatenate([t, q])


def read_kitti_calib_tr(path):
    """calib.txt, 5th line `Tr: r11 ... t3` -> (R, t) velodyne-to-camera (kitti_helper.cc:71-93)."""
    with open(path) as f:
        lines = f.read().splitlines()
    if len(lines) < 5 or not lines[4].startswith("Tr: "):
        raise ValueError("Tr parse failed!")                    # the reference exits here (:78-80)
    m = _f32_row(lines[4][4:]).reshape(3, 4)
    return m[:, :3], m[:, 3]


def read_kitti_times(path):
    with open(path) as f:
        return np.array([np.float32(l) for l in f.read().split()], dtype=np.float32)


def read_kitti_ground_truth(poses_path, Tr):
    """poses/NN.txt (camera frame) -> (n, 7) lidar-frame poses, Tl = Tr^-1 * Tc * Tr (kitti_helper.cc:99-119)."""
    Rr, tr = Tr
    Rr_inv = Rr.T                                              # Rigid3d::inverse(): quaternion conjugate
    out = []
    with open(poses_path) as f:
        for line in f.read().splitlines():
            if not line.strip():
                continue
            m = _f32_row(line).reshape(3, 4)
            Rc, tc = m[:, :3], m[:, 3]
            R = Rr_inv @ Rc @ Rr
            t = Rr_inv @ (Rc @ tr + tc - tr)
            p = _pose7(R, t)
            p[3:] /= np.linalg.norm(p[3:])                     # Tl.rotation().normalize() (:119)
            out.append(p)
    return np.array(out).reshape(-1, 7)


def rings_from_elevation(pts, n_scans=64, low_deg=-24.8, high_deg=2.0):
    """Ring index from the elevation angle, uniform bins over [low, high].  Returns (ring uint16,
    keep mask); points outside the fan are dropped the way a driver never produces them."""
    p = np.asarray(pts, dtype=np.float32)
    ang = np.degrees(np.arctan2(p[:, 2].astype(np.float64), np.hypot(p[:, 0].astype(np.float64), p[:, 1].astype(np.float64))))
    step = (high_deg - low_deg) / (n_scans - 1)
    r = np.floor((ang - low_deg) / step + 0.5).astype(np.int64)
    keep = (r >= 0) & (r < n_scans) & np.isfinite(ang)
    return np.clip(r, 0, n_scans - 1).astype(np.uint16), keep


class KittiSequence:
    """sequences/NN/{velodyne/*.bin, times.txt, calib.txt} + poses/NN.txt (optional)."""

    def __init__(self, dataset_folder, sequence_number):
        self.seq_dir = os.path.join(dataset_folder, "sequences", sequence_number)
        self.times = read_kitti_times(os.path.join(self.seq_dir, "times.txt"))
        self.Tr = read_kitti_calib_tr(os.path.join(self.seq_dir, "calib.txt"))
        gt = os.path.join(dataset_folder, "poses", sequence_number + ".txt")
        self.ground_truth = read_kitti_ground_truth(gt, self.Tr) if os.path.exists(gt) else None

    def __len__(self):
        return len(self.times)

    def scan(self, i, n_scans=64, low_deg=-24.8, high_deg=2.0):
        """-> (points (n,4) with t = 0, ring (n,) uint16); relative times are filled by extraction (A2)."""
        pts = read_kitti_bin(os.path.join(self.seq_dir, "velodyne", "%06d.bin" % i))
        ring, keep = rings_from_elevation(pts, n_scans, low_deg, high_deg)
        pts, ring = pts[keep], ring[keep]
        pts[:, 3] = 0.0
        return pts, ring


def write_kitti_sequence(dataset_folder, sequence_number, scans, times, poses_lidar=None, Tr=None):
    """Write scans (list of (n,4) arrays) in the layout above — test fixtures and synthetic replays."""
    seq_dir = os.path.join(dataset_folder, "sequences", sequence_number)
    os.makedirs(os.path.join(seq_dir, "velodyne"), exist_ok=True)
    for i, s in enumerate(scans):
        write_kitti_bin(os.path.join(seq_dir, "velodyne", "%06d.bin" % i), s)
    with open(os.path.join(seq_dir, "times.txt"), "w") as f:
        f.write("".join("%.6e\n" % t for t in times))
    R, t = Tr if Tr is not None else (np.eye(3), np.zeros(3))
    row = " ".join("%.9e" % v for v in np.hstack([R, t[:, None]]).reshape(-1))
    with open(os.path.join(seq_dir, "calib.txt"), "w") as f:
        f.write("P0: 0\nP1: 0\nP2: 0\nP3: 0\nTr: " + row + "\n")
    if poses_lidar is not None:
        from .synth import quat_to_matrix
        os.makedirs(os.path.join(dataset_folder, "poses"), exist_ok=True)
        with open(os.path.join(dataset_folder, "poses", sequence_number + ".txt"), "w") as f:
            for p in poses_lidar:
                Rl, tl = quat_to_matrix(p[3:]), p[:3]
                Rc = R @ Rl @ R.T                              # Tc = Tr * Tl * Tr^-1
                tc = R @ tl + t - Rc @ t
                f.write(" ".join("%.9e" % v for v in np.hstack([Rc, tc[:, None]]).reshape(-1)) + "\n")


# ------------------------------------------------------------------------------------ time

def from_seconds(seconds):
    """FromSeconds (common/time.cc:5-8): duration_cast truncates toward zero; ticks are nanoseconds."""
    return int(np.trunc(float(seconds) * 1e9))


def to_seconds(ticks):
    return float(ticks) / 1e9


# ------------------------------------------------------------------------------------ pose log (proto/msg.proto)

def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _doubles_msg(vals):
    """Vector3d / Quaterniond: fields 1.. as fixed64; proto3 omits fields whose value is +0.0."""
    out = bytearray()
    for i, v in enumerate(vals):
        bits = struct.pack("<d", float(v))
        if bits != b"\x00" * 8:
            out += bytes([((i + 1) << 3) | 1]) + bits
    return bytes(out)


def _len_field(field, payload):
    return bytes([(field << 3) | 2]) + _varint(len(payload)) + payload


def _read_varint(buf, i):
    v, shift = 0, 0
    while True:
        b = buf[i]
        i += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, i
        shift += 7


def _parse_fields(buf):
    i, out = 0, []
    while i < len(buf):
        key, i = _read_varint(buf, i)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _read_varint(buf, i)
        elif wt == 1:
            v = struct.unpack_from("<d", buf, i)[0]
            i += 8
        elif wt == 2:
            n, i = _read_varint(buf, i)
            v = bytes(buf[i:i + n])
            i += n
        elif wt == 5:
            v = struct.unpack_from("<f", buf, i)[0]
            i += 4
        else:
            raise ValueError("unsupported wire type %d" % wt)
        out.append((field, wt, v))
    return out


def _parse_doubles(buf, n):
    vals = [0.0] * n
    for field, wt, v in _parse_fields(buf):
        if wt == 1 and 1 <= field <= n:
            vals[field - 1] = v
    return vals


class PoseLog:
    """proto::PbData (msg.proto:34-37): odom_datas as LaserMapping appends them (laser_mapping.cc:251-254),
    imu_datas as AddImu does (:412-415)."""

    def __init__(self):
        self.odom = []     # (ticks, pose7)
        self.imu = []      # (ticks, acc[3], gyr[3])

    def add_odom(self, ticks, pose7):
        self.odom.append((int(ticks), np.array(pose7, dtype=np.float64)))

    def add_imu(self, ticks, linear_acceleration, angular_velocity):
        self.imu.append((int(ticks), np.array(linear_acceleration, np.float64), np.array(angular_velocity, np.float64)))

    def serialize(self):
        out = bytearray()
        for ticks, acc, gyr in self.imu:                                  # field 1 first: protobuf writes in field order
            m = (bytes([0x08]) + _varint(ticks) if ticks else b"") + _len_field(2, _doubles_msg(acc)) + _len_field(3, _doubles_msg(gyr))
            out += _len_field(1, m)
        for ticks, p in self.odom:
            rigid = _len_field(1, _doubles_msg(p[:3])) + _len_field(2, _doubles_msg(p[3:7]))
            m = (bytes([0x08]) + _varint(ticks) if ticks else b"") + _len_field(2, rigid)
            out += _len_field(2, m)
        return bytes(out)

    @classmethod
    def parse(cls, data):
        log = cls()
        for field, wt, v in _parse_fields(data):
            if wt != 2:
                continue
            ticks, sub = 0, {}
            for f2, w2, v2 in _parse_fields(v):
                if f2 == 1 and w2 == 0:
                    ticks = v2 if v2 < (1 << 63) else v2 - (1 << 64)
                elif w2 == 2:
                    sub[f2] = v2
            if field == 2:
                t, q = [0.0] * 3, [0.0] * 4
                for f3, w3, v3 in _parse_fields(sub.get(2, b"")):
                    if w3 == 2 and f3 == 1:
                        t = _parse_doubles(v3, 3)
                    elif w3 == 2 and f3 == 2:
                        q = _parse_doubles(v3, 4)
                log.add_odom(ticks, t + q)
            elif field == 1:
                log.add_imu(ticks, _parse_doubles(sub.get(2, b""), 3), _parse_doubles(sub.get(3, b""), 3))
        return log

    def save(self, path):
        with open(path, "wb") as f:
            f.write(self.serialize())

    @classmethod
    def load(cls, path):
        with open(path, "rb") as f:
            return cls.parse(f.read())


# ------------------------------------------------------------------------------------ ATE

def ate_rmse(est, truth, align=True):
    """Absolute trajectory error: RMSE of positions, after a rigid (Horn/Kabsch, no scale) alignment
    of est onto truth when `align`."""
    e = np.asarray(est, dtype=np.float64)[:, :3]
    g = np.asarray(truth, dtype=np.float64)[:, :3]
    if align and len(e) >= 3:
        ce, cg = e.mean(0), g.mean(0)
        H = (e - ce).T @ (g - cg)
        U, _, Vt = np.linalg.svd(H)
        D = np.diag([1.0, 1.0, np.sign(np.linalg.det(Vt.T @ U.T))])
        R = Vt.T @ D @ U.T
        e = (e - ce) @ R.T + cg
    return float(np.sqrt(np.mean(np.sum((e - g) ** 2, axis=1))))
