"""Synthetic plane+pole world, VLP-16 / 64-beam ray-cast scans and sampled local maps.

Harness only (inputs for tests and bench.py; SURVEY.md §8d recipe, seed 20260928).  Nothing here
is on the product path and nothing here touches oracle/.

World: ground plane z=0 (|x|,|y| <= ground_half), a 60 x 40 m room (walls x=+-30, y=+-20,
height 6 m), `n_poles` vertical poles (r=0.1 m, h=4 m) at seeded positions inside the room.
Sensor: 16 beams -15..+15 deg step 2 deg x 1800 azimuth steps (0.2 deg), clockwise azimuth
(matches the reference's `-atan2(y, x)` convention, msf_loam_node.cc:131-139), driver order =
azimuth-major / ring-minor, ring = beam index by ascending elevation, range noise N(0, 0.01 m),
returns outside [0.3, 100] m dropped.
"""
import numpy as np

SEED = 20260928
ROOM_HX, ROOM_HY, WALL_H = 30.0, 20.0, 6.0
POLE_R, POLE_H = 0.1, 4.0


# ---- quaternion helpers ([x y z w], Hamilton, like Eigen) --------------------------------------

def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz])


def quat_from_rotvec(v):
    v = np.asarray(v, dtype=np.float64)
    th = np.linalg.norm(v)
    if th < 1e-12:
        return np.array([0.5 * v[0], 0.5 * v[1], 0.5 * v[2], 1.0])
    s = np.sin(th / 2) / th
    return np.array([s * v[0], s * v[1], s * v[2], np.cos(th / 2)])


def quat_to_matrix(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def quat_from_euler(roll, pitch, yaw):
    qx = quat_from_rotvec([roll, 0, 0])
    qy = quat_from_rotvec([0, pitch, 0])
    qz = quat_from_rotvec([0, 0, yaw])
    q = quat_mul(qz, quat_mul(qy, qx))
    return q / np.linalg.norm(q)


def pose_error(a, b):
    """(translation distance [m], rotation angle [rad]) between two Vector7 poses."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    dt = float(np.linalg.norm(a[:3] - b[:3]))
    qa = a[3:] / np.linalg.norm(a[3:])
    qb = b[3:] / np.linalg.norm(b[3:])
    d = abs(float(np.dot(qa, qb)))
    # angle = 2 acos(|<qa,qb>|); use the sine form for accuracy near zero
    qc = np.array([-qa[0], -qa[1], -qa[2], qa[3]])
    rel = quat_mul(qc, qb)
    ang = 2.0 * float(np.arctan2(np.linalg.norm(rel[:3]), abs(rel[3])))
    del d
    return dt, ang


# ---- world ------------------------------------------------------------------------------------

class World:
    """kind = "room" (ground plane, 60 x 40 m room, poles), "outdoor" or "corridor" (round 5, `worlds.py`:
    relief + buildings + trunks + volumetric canopy; a featureless 80 m corridor).  The room's streams of random numbers are
    untouched by the other kinds: every seeded room case of rounds 1-4 is bit-identical."""

    def __init__(self, seed=SEED, n_poles=12, ground_half=80.0, kind="room", **kw):
        rng = np.random.default_rng(seed)
        self.kind = kind
        self.ground_half = float(ground_half)
        self.poles = np.stack([rng.uniform(-ROOM_HX + 2, ROOM_HX - 2, n_poles),
                               rng.uniform(-ROOM_HY + 2, ROOM_HY - 2, n_poles)], axis=1)
        self.seed = seed
        self.geom = None
        if kind == "outdoor":
            from . import worlds
            self.geom = worlds.Outdoor(seed + 11, **kw)
        elif kind == "corridor":
            from . import worlds
            self.geom = worlds.Corridor(seed + 12, **kw)
        elif kind != "room":
            raise ValueError("unknown world kind %r" % (kind,))


def world_poses(world, n, seed=SEED + 2):
    """n sensor poses (scan -> world) suited to the world: `random_poses` in the room, the geometry's own elsewhere."""
    return random_poses(n, seed) if world.geom is None else world.geom.random_poses(n, seed)


def ground_half_for_target(total_points, n_poles=12, surf_spacing=0.4, corner_spacing=0.2):
    """Ground half-extent so the sampled map has ~total_points points."""
    walls = 2 * (2 * ROOM_HX + 2 * ROOM_HY) * WALL_H / (surf_spacing ** 2) / 2
    corners = (n_poles * POLE_H + 4 * WALL_H + 2 * (2 * ROOM_HX + 2 * ROOM_HY)) / corner_spacing
    ground = max(total_points - walls - corners, 1000.0)
    return 0.5 * np.sqrt(ground) * surf_spacing


def make_map(world, seed=SEED + 1, corner_spacing=0.2, surf_spacing=0.4, jitter=0.005):
    """Local map clouds sampled from the world geometry: (corner (mc,4), surf (ms,4)) float32."""
    rng = np.random.default_rng(seed)
    if world.geom is not None:
        corner, surf = world.geom.make_map(rng, corner_spacing, surf_spacing, jitter)
        surf = surf[rng.permutation(len(surf))]
        corner = corner[rng.permutation(len(corner))]
        out = []
        for a in (corner, surf):
            p = np.zeros((len(a), 4), np.float32)
            p[:, :3] = a.astype(np.float32)
            out.append(p)
        return out[0], out[1]
    g = world.ground_half
    ax = np.arange(-g, g + 1e-9, surf_spacing)
    gx, gy = np.meshgrid(ax, ax, indexing="ij")
    surf = [np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], axis=1)]
    zs = np.arange(surf_spacing / 2, WALL_H, surf_spacing)
    xs = np.arange(-ROOM_HX, ROOM_HX + 1e-9, surf_spacing)
    ys = np.arange(-ROOM_HY, ROOM_HY + 1e-9, surf_spacing)
    for sx in (-ROOM_HX, ROOM_HX):
        yy, zz = np.meshgrid(ys, zs, indexing="ij")
        surf.append(np.stack([np.full(yy.size, sx), yy.ravel(), zz.ravel()], axis=1))
    for sy in (-ROOM_HY, ROOM_HY):
        xx, zz = np.meshgrid(xs, zs, indexing="ij")
        surf.append(np.stack([xx.ravel(), np.full(xx.size, sy), zz.ravel()], axis=1))
    surf = np.concatenate(surf)
    corner = []
    zc = np.arange(0.0, POLE_H + 1e-9, corner_spacing)
    for px, py in world.poles:
        corner.append(np.stack([np.full(zc.size, px), np.full(zc.size, py), zc], axis=1))
    zw = np.arange(0.0, WALL_H + 1e-9, corner_spacing)
    for sx in (-ROOM_HX, ROOM_HX):
        for sy in (-ROOM_HY, ROOM_HY):
            corner.append(np.stack([np.full(zw.size, sx), np.full(zw.size, sy), zw], axis=1))
    xe = np.arange(-ROOM_HX, ROOM_HX + 1e-9, corner_spacing)
    ye = np.arange(-ROOM_HY, ROOM_HY + 1e-9, corner_spacing)
    for sy in (-ROOM_HY, ROOM_HY):
        corner.append(np.stack([xe, np.full(xe.size, sy), np.zeros(xe.size)], axis=1))
    for sx in (-ROOM_HX, ROOM_HX):
        corner.append(np.stack([np.full(ye.size, sx), ye, np.zeros(ye.size)], axis=1))
    corner = np.concatenate(corner)
    surf = surf + rng.normal(0, jitter, surf.shape)
    corner = corner + rng.normal(0, jitter, corner.shape)
    # a real map has no particular order: shuffle so nothing downstream can rely on it
    surf = surf[rng.permutation(len(surf))]
    corner = corner[rng.permutation(len(corner))]
    out = []
    for a in (corner, surf):
        p = np.zeros((len(a), 4), np.float32)
        p[:, :3] = a.astype(np.float32)
        out.append(p)
    return out[0], out[1]


# ---- sensor -----------------------------------------------------------------------------------

def beam_dirs(n_beams=16, n_az=1800, elev_lo=-15.0, elev_hi=15.0):
    """Unit ray directions in the sensor frame, driver order (azimuth-major, ring-minor)."""
    elev = np.deg2rad(np.linspace(elev_lo, elev_hi, n_beams))
    az = -2.0 * np.pi * np.arange(n_az) / n_az          # clockwise
    ce, se = np.cos(elev), np.sin(elev)
    d = np.stack([np.outer(np.cos(az), ce), np.outer(np.sin(az), ce), np.outer(np.ones(n_az), se)], axis=2)
    ring = np.tile(np.arange(n_beams, dtype=np.uint16), n_az)
    return d.reshape(-1, 3), ring


def raycast(world, R, t, dirs_local, rng=None):
    """Range and hit kind (0 ground, 1 wall, 2 pole, 3 volume, -1 miss) of rays from t along R @ d.  `rng`: only the
    volumetric returns of the outdoor world draw from it."""
    d = dirs_local @ R.T
    o = np.asarray(t, dtype=np.float64)
    if world.geom is not None:
        return world.geom.raycast(o, d, rng if rng is not None else np.random.default_rng(0))
    n = len(d)
    best = np.full(n, np.inf)
    kind = np.full(n, -1, np.int8)
    with np.errstate(divide="ignore", invalid="ignore"):
        # ground
        s = -o[2] / d[:, 2]
        hx, hy = o[0] + s * d[:, 0], o[1] + s * d[:, 1]
        ok = (d[:, 2] < 0) & (s > 0) & (np.abs(hx) <= world.ground_half) & (np.abs(hy) <= world.ground_half)
        best = np.where(ok, s, best)
        kind = np.where(ok, 0, kind)
        # walls
        for axis, half, other_half in ((0, ROOM_HX, ROOM_HY), (1, ROOM_HY, ROOM_HX)):
            for sgn in (-1.0, 1.0):
                s = (sgn * half - o[axis]) / d[:, axis]
                ho = o[1 - axis] + s * d[:, 1 - axis]
                hz = o[2] + s * d[:, 2]
                ok = (s > 0) & (np.abs(ho) <= other_half) & (hz >= 0) & (hz <= WALL_H) & (s < best)
                best = np.where(ok, s, best)
                kind = np.where(ok, 1, kind)
        # poles (vertical cylinders)
        a = d[:, 0] ** 2 + d[:, 1] ** 2
        for px, py in world.poles:
            ox, oy = o[0] - px, o[1] - py
            b = ox * d[:, 0] + oy * d[:, 1]
            c = ox * ox + oy * oy - POLE_R ** 2
            disc = b * b - a * c
            s = (-b - np.sqrt(np.maximum(disc, 0))) / a
            hz = o[2] + s * d[:, 2]
            ok = (disc > 0) & (s > 0) & (hz >= 0) & (hz <= POLE_H) & (s < best)
            best = np.where(ok, s, best)
            kind = np.where(ok, 2, kind)
    return best, kind


def make_scan(world, pose, seed, n_beams=16, n_az=1800, elev=(-15.0, 15.0), noise=0.01, with_kind=False):
    """One sensor cloud in the sensor frame: pts (n,4) f32 (t=0), ring (n,) u16, driver order."""
    rng = np.random.default_rng(seed)
    dirs, ring = beam_dirs(n_beams, n_az, elev[0], elev[1])
    R = quat_to_matrix(np.asarray(pose[3:], dtype=np.float64))
    rng_m, kind = raycast(world, R, pose[:3], dirs, rng)
    rng_m = rng_m + rng.normal(0, noise, rng_m.shape)
    ok = np.isfinite(rng_m) & (rng_m >= 0.3) & (rng_m <= 100.0) & (kind >= 0)
    p = dirs[ok] * rng_m[ok, None]
    pts = np.zeros((len(p), 4), np.float32)
    pts[:, :3] = p.astype(np.float32)
    if with_kind:
        return pts, ring[ok].copy(), kind[ok].copy()
    return pts, ring[ok].copy()


def random_poses(n, seed=SEED + 2):
    """Sensor poses (scan->world): t ~ U(-10,10)^2 x (1.8 + U(-0.2,0.2)), yaw U(-pi,pi),
    roll/pitch N(0, 2 deg)."""
    rng = np.random.default_rng(seed)
    poses = np.zeros((n, 7))
    for i in range(n):
        poses[i, 0:2] = rng.uniform(-10, 10, 2)
        poses[i, 2] = 1.8 + rng.uniform(-0.2, 0.2)
        roll, pitch = rng.normal(0, np.deg2rad(2.0), 2)
        poses[i, 3:] = quat_from_euler(roll, pitch, rng.uniform(-np.pi, np.pi))
    return poses


def perturb_pose(pose, rng, max_t=0.3, max_deg=3.0):
    """truth (+) perturbation with |dt| <= max_t, |dtheta| <= max_deg (right-multiplied)."""
    dt = rng.normal(size=3)
    dt *= rng.uniform(0, max_t) / np.linalg.norm(dt)
    dr = rng.normal(size=3)
    dr *= np.deg2rad(rng.uniform(0, max_deg)) / np.linalg.norm(dr)
    out = np.array(pose, dtype=np.float64)
    out[:3] += dt
    q = quat_mul(out[3:], quat_from_rotvec(dr))
    out[3:] = q / np.linalg.norm(q)
    return out


def voxel_downsample_np(pts, leaf):
    """numpy voxel-centroid filter (pcl::VoxelGrid semantics, f32 arithmetic); harness helper."""
    pts = np.asarray(pts, dtype=np.float32).reshape(-1, 4)
    if len(pts) == 0:
        return pts.copy()
    inv = np.float32(1.0) / np.float32(leaf)
    ijk = np.floor(pts[:, :3] * inv).astype(np.int64)
    mn = ijk.min(axis=0)
    div = ijk.max(axis=0) - mn + 1
    key = (ijk[:, 0] - mn[0]) + (ijk[:, 1] - mn[1]) * div[0] + (ijk[:, 2] - mn[2]) * div[0] * div[1]
    order = np.argsort(key, kind="stable")
    key_s = key[order]
    starts = np.flatnonzero(np.r_[True, key_s[1:] != key_s[:-1]])
    sums = np.add.reduceat(pts[order].astype(np.float64), starts, axis=0)
    cnt = np.diff(np.r_[starts, len(pts)])[:, None]
    return (sums / cnt).astype(np.float32)


def direct_features(pts, kind, corner_leaf=0.2, surf_leaf=0.4):
    """Feature clouds straight from the ray-cast hit kinds (pole hits -> corner, plane hits ->
    surf), voxel down-sampled like laser_mapping.cc:264-270.  Used only until the product's own
    extraction feeds the registration bench."""
    corner = voxel_downsample_np(pts[kind == 2], corner_leaf)
    surf = voxel_downsample_np(pts[kind != 2], surf_leaf)
    return corner, surf
