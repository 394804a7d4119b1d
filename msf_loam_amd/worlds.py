"""Two more synthetic worlds for the harness (round 5): `outdoor` and `corridor`.

`synth.World(kind="room")` is a ground plane, four walls and 12 poles: surfaces at the map's leaf spacing, ~24
5-NN candidates per query, always well constrained.  It is the friendliest case for a grid 5-NN and for the LM solve, and the
closest thing to BASELINE configs[2]'s `nsh_indoor_outdoor.bag` (README.md:46) that can exist offline is a world with what
that drive has and the room lacks:

  outdoor   200 x 200 m: ground with +-0.3 m relief, 40 oriented box "buildings", 150 trunks (cylinders), a canopy ball on
            every trunk and 60 bushes on the ground.  Canopy and bushes are VOLUMES: a ray that enters one returns from a random
            depth (exponential free path) or passes through, and the sampled local map fills them at the 0.4 m / 0.2 m leaf
            spacing of the map's voxel filters (laser_mapping.cc:60-68), i.e. up to 15.6 / 125 points per m^3 — hundreds of
            candidates in the 27 cells of a query instead of two dozen, features whose five neighbours fit no line / plane.
  corridor  80 x 3 x 3 m box, optionally a few shallow pilasters on the walls.  Nothing but the two end walls (seen by two rings
            at 40 m) constrains the motion along the axis: the normal equations are nearly singular, Levenberg-Marquardt
            steps get rejected and the trust region shrinks; few corner features, so `LaserMapping`'s map gate
            (laser_mapping.cc:284-285) can close.

Harness only: inputs for tests and bench.py.  Nothing here is on the product path and nothing here touches oracle/.
Hit kinds: 0 ground / floor, 1 wall (building face, corridor wall, ceiling), 2 pole / trunk / pilaster edge region, 3 volume.
"""
import numpy as np

OUT_HALF = 100.0            # outdoor: |x|, |y| <= 100 m
COR_HX, COR_HY, COR_H = 40.0, 1.5, 3.0


def _rot2(yaw):
    c, s = np.cos(yaw), np.sin(yaw)
    return c, s


# ---- outdoor ---------------------------------------------------------------------------------------------------------

class Outdoor:
    def __init__(self, seed, n_buildings=40, n_trunks=150, n_bushes=60, lane_half=3.0):
        rng = np.random.default_rng(seed)
        self.half = OUT_HALF
        # relief: three sinusoids, amplitude sum 0.3 m, wavelengths 37 / 23 / 11 m
        self.relief = [(0.15, 2 * np.pi / 37.0, 0.0, rng.uniform(0, 2 * np.pi)),
                       (0.10, 0.0, 2 * np.pi / 23.0, rng.uniform(0, 2 * np.pi)),
                       (0.05, 2 * np.pi / 11.0 / np.sqrt(2), 2 * np.pi / 11.0 / np.sqrt(2), rng.uniform(0, 2 * np.pi))]
        # buildings: oriented boxes (cx, cy, hx, hy, yaw, height); none across the two "streets" x ~ 0 and y ~ 0, where the
        # sensor drives / the random poses are drawn
        b = []
        while len(b) < n_buildings:
            cx, cy = rng.uniform(-90, 90, 2)
            hx, hy = rng.uniform(4, 12, 2)
            yaw = rng.uniform(-0.4, 0.4)
            r = np.hypot(hx, hy)
            if abs(cx) < r + lane_half or abs(cy) < r + lane_half:
                continue
            if any(np.hypot(cx - o[0], cy - o[1]) < r + np.hypot(o[2], o[3]) + 1.0 for o in b):
                continue
            b.append((cx, cy, hx, hy, yaw, rng.uniform(4, 12)))
        self.buildings = np.array(b)
        # trunks (px, py, r, h) outside the buildings, canopy ball (cx, cy, cz, r) on each; bushes are balls on the ground
        t = []
        while len(t) < n_trunks:
            px, py = rng.uniform(-95, 95, 2)
            if self.inside_building(px, py, 1.0) or (abs(px) < 1.5 and abs(py) < 1.5):
                continue
            t.append((px, py, rng.uniform(0.1, 0.3), rng.uniform(3.0, 6.0)))
        self.trunks = np.array(t)
        cr = rng.uniform(2.0, 3.5, n_trunks)
        self.balls = [(t[i][0], t[i][1], self.ground(t[i][0], t[i][1]) + t[i][3] + 0.5 * cr[i], cr[i]) for i in range(n_trunks)]
        k = 0
        while k < n_bushes:
            px, py = rng.uniform(-95, 95, 2)
            if self.inside_building(px, py, 1.5) or np.hypot(px, py) < 3.0:
                continue
            r = rng.uniform(0.8, 1.5)
            self.balls.append((px, py, self.ground(px, py) + 0.4 * r, r))
            k += 1
        self.balls = np.array(self.balls)
        self.free_path = 0.8            # mean depth [m] a ray travels inside a volume before it returns

    def ground(self, x, y):
        z = 0.0
        for a, kx, ky, ph in self.relief:
            z = z + a * np.sin(kx * x + ky * y + ph)
        return z

    def inside_building(self, x, y, margin=0.0):
        if not hasattr(self, "buildings") or len(self.buildings) == 0:
            return False
        for cx, cy, hx, hy, yaw, _ in self.buildings:
            c, s = _rot2(yaw)
            u, v = c * (x - cx) + s * (y - cy), -s * (x - cx) + c * (y - cy)
            if abs(u) < hx + margin and abs(v) < hy + margin:
                return True
        return False

    def inside_volume(self, x, y, z, margin=0.0):
        b = self.balls
        return bool(np.any((b[:, 0] - x) ** 2 + (b[:, 1] - y) ** 2 + (b[:, 2] - z) ** 2 < (b[:, 3] + margin) ** 2))

    # -- sensor side
    def raycast(self, o, d, rng):
        n = len(d)
        best = np.full(n, np.inf)
        kind = np.full(n, -1, np.int8)
        with np.errstate(divide="ignore", invalid="ignore"):
            # ground: only rays that go down can meet a +-0.3 m relief from 1.8 m; the crossing lies between the planes z = +0.35 and
            # z = -0.35: 12 samples of that stretch find the first sign change, six bisections and a secant step refine it
            dn = np.flatnonzero(d[:, 2] < -1e-6)
            if len(dn):
                dd = d[dn]
                s0 = np.maximum((0.35 - o[2]) / dd[:, 2], 0.0)
                s1 = (-0.35 - o[2]) / dd[:, 2]
                f = lambda s: o[2] + s * dd[:, 2] - self.ground(o[0] + s * dd[:, 0], o[1] + s * dd[:, 1])   # noqa: E731
                lo, hi = s0.copy(), s1.copy()
                found = np.zeros(len(dn), bool)
                prev = s0
                for k in range(1, 13):
                    s = s0 + (s1 - s0) * (k / 12.0)
                    neg = (f(s) <= 0) & ~found
                    lo = np.where(neg, prev, lo); hi = np.where(neg, s, hi)
                    found |= neg
                    prev = s
                for _ in range(6):
                    mid = 0.5 * (lo + hi)
                    neg = f(mid) <= 0
                    hi = np.where(neg, mid, hi); lo = np.where(neg, lo, mid)
                flo, fhi = f(lo), f(hi)                                      # one secant step inside the final bracket
                s = lo + (hi - lo) * np.clip(flo / np.where(flo - fhi != 0, flo - fhi, 1.0), 0.0, 1.0)
                hx, hy = o[0] + s * dd[:, 0], o[1] + s * dd[:, 1]
                ok = found & (np.abs(hx) <= self.half) & (np.abs(hy) <= self.half)
                best[dn] = np.where(ok, s, np.inf)
                kind[dn] = np.where(ok, 0, -1)
            # everything else subtends a small horizontal angle: rays are binned by azimuth (0.5 degree bins) once and each primitive only
            # looks at the bins its bounding circle covers
            nb = 720
            az_bin = np.floor((np.arctan2(d[:, 1], d[:, 0]) + np.pi) * (nb / (2 * np.pi))).astype(np.int64) % nb
            steep = d[:, 0] ** 2 + d[:, 1] ** 2 < 1e-6                     # (near-)vertical rays belong to every bin: none with these sensors
            order = np.argsort(az_bin, kind="stable")
            starts = np.searchsorted(az_bin[order], np.arange(nb + 1))

            def sector(cx, cy, r):
                ox, oy = cx - o[0], cy - o[1]
                L = np.hypot(ox, oy)
                if L <= r * 1.05 or steep.any():
                    return order
                half = np.arcsin(r / L) + 2 * np.pi / nb
                c = np.arctan2(oy, ox) + np.pi
                b0 = int(np.floor((c - half) * (nb / (2 * np.pi)))); b1 = int(np.floor((c + half) * (nb / (2 * np.pi))))
                if b1 - b0 >= nb - 1:
                    return order
                b0m, b1m = b0 % nb, b1 % nb
                if b0m <= b1m:
                    return order[starts[b0m]:starts[b1m + 1]]
                return np.concatenate([order[starts[b0m]:], order[:starts[b1m + 1]]])

            # buildings: slab test in the box frame (z from -1 m, below the relief, to the roof)
            for cx, cy, bx, by, yaw, h in self.buildings:
                idx = sector(cx, cy, np.hypot(bx, by))
                if not len(idx):
                    continue
                di = d[idx]
                c, s_ = _rot2(yaw)
                ox, oy = c * (o[0] - cx) + s_ * (o[1] - cy), -s_ * (o[0] - cx) + c * (o[1] - cy)
                dx, dy = c * di[:, 0] + s_ * di[:, 1], -s_ * di[:, 0] + c * di[:, 1]
                tx0, tx1 = (-bx - ox) / dx, (bx - ox) / dx
                ty0, ty1 = (-by - oy) / dy, (by - oy) / dy
                tz0, tz1 = (-1.0 - o[2]) / di[:, 2], (h - o[2]) / di[:, 2]
                t_in = np.maximum(np.maximum(np.minimum(tx0, tx1), np.minimum(ty0, ty1)), np.minimum(tz0, tz1))
                t_out = np.minimum(np.minimum(np.maximum(tx0, tx1), np.maximum(ty0, ty1)), np.maximum(tz0, tz1))
                ok = (t_in < t_out) & (t_in > 0) & (t_in < best[idx])
                best[idx[ok]] = t_in[ok]
                kind[idx[ok]] = 1
            # trunks
            for px, py, r, h in self.trunks:
                ox, oy = o[0] - px, o[1] - py
                if ox * ox + oy * oy > 110.0 ** 2:
                    continue
                idx = sector(px, py, r)
                if not len(idx):
                    continue
                di = d[idx]
                a = di[:, 0] ** 2 + di[:, 1] ** 2
                b = ox * di[:, 0] + oy * di[:, 1]
                disc = b * b - a * (ox * ox + oy * oy - r * r)
                s = (-b - np.sqrt(np.maximum(disc, 0))) / a
                hz = o[2] + s * di[:, 2]
                ok = (disc > 0) & (s > 0) & (hz >= -1.0) & (hz <= h + 0.3) & (s < best[idx])     # the relief is within +-0.3 m of z = 0
                best[idx[ok]] = s[ok]
                kind[idx[ok]] = 2
            # volumes: chord [s_in, s_out] of the ball, return from s_in + Exp(free_path) if that is still inside
            for cx, cy, cz, r in self.balls:
                oc = np.array([o[0] - cx, o[1] - cy, o[2] - cz])
                if np.dot(oc, oc) > (105.0 + r) ** 2:
                    continue
                idx = sector(cx, cy, r)
                if not len(idx):
                    continue
                di = d[idx]
                b = di @ oc
                disc = b * b - (np.dot(oc, oc) - r * r)
                sq = np.sqrt(np.maximum(disc, 0))
                s_in, s_out = np.maximum(-b - sq, 0.0), -b + sq
                s = s_in + rng.exponential(self.free_path, len(idx))
                ok = (disc > 0) & (s_out > 0) & (s < s_out) & (s < best[idx])
                best[idx[ok]] = s[ok]
                kind[idx[ok]] = 3
        return best, kind

    def random_poses(self, n, seed):
        """Sensor poses: anywhere within |x|, |y| <= 60 m that is outside the buildings (1.5 m clearance) and the volumes."""
        from . import synth
        rng = np.random.default_rng(seed)
        poses = np.zeros((n, 7))
        for i in range(n):
            while True:
                x, y = rng.uniform(-60, 60, 2)
                z = self.ground(x, y) + 1.8 + rng.uniform(-0.2, 0.2)
                if not self.inside_building(x, y, 1.5) and not self.inside_volume(x, y, z, 0.5) and \
                        np.min(np.hypot(self.trunks[:, 0] - x, self.trunks[:, 1] - y)) > 0.8:
                    break
            roll, pitch = rng.normal(0, np.deg2rad(2.0), 2)
            poses[i, :3] = x, y, z
            poses[i, 3:] = synth.quat_from_euler(roll, pitch, rng.uniform(-np.pi, np.pi))
        return poses

    # -- map side
    def make_map(self, rng, corner_spacing, surf_spacing, jitter, fill_surf=0.75, fill_corner=0.10):
        from . import synth
        sp = surf_spacing
        ax = np.arange(-self.half, self.half + 1e-9, sp)
        gx, gy = np.meshgrid(ax, ax, indexing="ij")
        gx, gy = gx.ravel(), gy.ravel()
        keep = np.ones(len(gx), bool)
        for cx, cy, bx, by, yaw, _ in self.buildings:
            c, s = _rot2(yaw)
            u, v = c * (gx - cx) + s * (gy - cy), -s * (gx - cx) + c * (gy - cy)
            keep &= ~((np.abs(u) < bx) & (np.abs(v) < by))
        gx, gy = gx[keep], gy[keep]
        surf = [np.stack([gx, gy, self.ground(gx, gy)], axis=1)]
        corner = []
        for cx, cy, bx, by, yaw, h in self.buildings:
            c, s = _rot2(yaw)
            zs = np.arange(sp / 2, h, sp)
            for (u0, v0, u1, v1) in ((-bx, -by, bx, -by), (bx, -by, bx, by), (bx, by, -bx, by), (-bx, by, -bx, -by)):
                L = np.hypot(u1 - u0, v1 - v0)
                tt = np.arange(0, L + 1e-9, sp) / L
                uu, vv = u0 + (u1 - u0) * tt, v0 + (v1 - v0) * tt
                xx, yy = cx + c * uu - s * vv, cy + s * uu + c * vv
                X, Z = np.meshgrid(xx, zs, indexing="ij"); Y, _ = np.meshgrid(yy, zs, indexing="ij")
                surf.append(np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1))
                # wall-ground edge and roof edge
                te = np.arange(0, L + 1e-9, corner_spacing) / L
                ue, ve = u0 + (u1 - u0) * te, v0 + (v1 - v0) * te
                xe, ye = cx + c * ue - s * ve, cy + s * ue + c * ve
                corner.append(np.stack([xe, ye, self.ground(xe, ye)], axis=1))
                corner.append(np.stack([xe, ye, np.full(len(xe), h)], axis=1))
                # the vertical edge at (u0, v0)
                zc = np.arange(0.0, h + 1e-9, corner_spacing)
                x0, y0 = cx + c * u0 - s * v0, cy + s * u0 + c * v0
                corner.append(np.stack([np.full(len(zc), x0), np.full(len(zc), y0), zc], axis=1))
        for px, py, r, h in self.trunks:
            zc = np.arange(0.0, h + 1e-9, corner_spacing) + self.ground(px, py)
            corner.append(np.stack([np.full(len(zc), px), np.full(len(zc), py), zc], axis=1))
        # volumes: a jittered lattice at the leaf spacing, a fraction of its nodes occupied
        vol_s, vol_c = [], []
        for cx, cy, cz, r in self.balls:
            for spacing, fill, out in ((sp, fill_surf, vol_s), (corner_spacing, fill_corner, vol_c)):
                a1 = np.arange(-r, r + 1e-9, spacing)
                X, Y, Z = np.meshgrid(a1, a1, a1, indexing="ij")
                p = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
                p = p[np.sum(p * p, axis=1) <= r * r]
                p = p[rng.uniform(size=len(p)) < fill]
                out.append(p + [cx, cy, cz] + rng.uniform(-0.3 * spacing, 0.3 * spacing, p.shape))
        surf = np.concatenate(surf) + rng.normal(0, jitter, (sum(len(a) for a in surf), 3))
        corner = np.concatenate(corner) + rng.normal(0, jitter, (sum(len(a) for a in corner), 3))
        vs, vc = np.concatenate(vol_s), np.concatenate(vol_c)
        # overlapping balls: one point per leaf voxel, like the map store's voxel filter would leave
        def pack(a):
            p = np.zeros((len(a), 4), np.float32); p[:, :3] = a
            return p
        vs = synth.voxel_downsample_np(pack(vs), sp)[:, :3]
        vc = synth.voxel_downsample_np(pack(vc), corner_spacing)[:, :3]
        self.n_volume_surf, self.n_volume_corner = len(vs), len(vc)
        return np.concatenate([corner, vc]), np.concatenate([surf, vs])


# ---- corridor --------------------------------------------------------------------------------------------------------

class Corridor:
    def __init__(self, seed, n_pilasters=2):
        rng = np.random.default_rng(seed)
        # pilasters: boxes 0.3 m along x, 0.12 m deep, full height, on either wall: (x0, x1, y0, y1)
        p = []
        for _ in range(n_pilasters):
            x0 = rng.uniform(-COR_HX + 5, COR_HX - 5)
            side = 1.0 if rng.uniform() < 0.5 else -1.0
            y_in = side * (COR_HY - 0.12)
            p.append((x0, x0 + 0.3, min(y_in, side * COR_HY), max(y_in, side * COR_HY)))
        self.pilasters = np.array(p).reshape(-1, 4)

    def raycast(self, o, d, rng):
        n = len(d)
        best = np.full(n, np.inf)
        kind = np.full(n, -1, np.int8)
        with np.errstate(divide="ignore", invalid="ignore"):
            def plane(axis, coord, k):
                nonlocal best, kind
                s = (coord - o[axis]) / d[:, axis]
                h = o[None, :] + s[:, None] * d
                ok = (s > 0) & (s < best)
                for a, half_lo, half_hi in ((0, -COR_HX, COR_HX), (1, -COR_HY, COR_HY), (2, 0.0, COR_H)):
                    if a != axis:
                        ok &= (h[:, a] >= half_lo - 1e-9) & (h[:, a] <= half_hi + 1e-9)
                best = np.where(ok, s, best)
                kind = np.where(ok, k, kind)
            plane(2, 0.0, 0)
            plane(2, COR_H, 1)
            plane(1, -COR_HY, 1); plane(1, COR_HY, 1)
            plane(0, -COR_HX, 1); plane(0, COR_HX, 1)
            for x0, x1, y0, y1 in self.pilasters:
                tx0, tx1 = (x0 - o[0]) / d[:, 0], (x1 - o[0]) / d[:, 0]
                ty0, ty1 = (y0 - o[1]) / d[:, 1], (y1 - o[1]) / d[:, 1]
                tz0, tz1 = (0.0 - o[2]) / d[:, 2], (COR_H - o[2]) / d[:, 2]
                t_in = np.maximum(np.maximum(np.minimum(tx0, tx1), np.minimum(ty0, ty1)), np.minimum(tz0, tz1))
                t_out = np.minimum(np.minimum(np.maximum(tx0, tx1), np.maximum(ty0, ty1)), np.maximum(tz0, tz1))
                ok = (t_in < t_out) & (t_in > 0) & (t_in < best)
                best = np.where(ok, t_in, best)
                kind = np.where(ok, 2, kind)
        return best, kind

    def random_poses(self, n, seed):
        from . import synth
        rng = np.random.default_rng(seed)
        poses = np.zeros((n, 7))
        for i in range(n):
            poses[i, 0] = rng.uniform(-30, 30)
            poses[i, 1] = rng.uniform(-0.6, 0.6)
            poses[i, 2] = 1.5 + rng.uniform(-0.2, 0.2)
            roll, pitch = rng.normal(0, np.deg2rad(2.0), 2)
            poses[i, 3:] = synth.quat_from_euler(roll, pitch, rng.uniform(-np.pi, np.pi))
        return poses

    def make_map(self, rng, corner_spacing, surf_spacing, jitter):
        sp = surf_spacing
        xs = np.arange(-COR_HX, COR_HX + 1e-9, sp)
        ys = np.arange(-COR_HY + sp / 2, COR_HY, sp)
        zs = np.arange(sp / 2, COR_H, sp)
        surf = []
        X, Y = np.meshgrid(xs, ys, indexing="ij")
        for z in (0.0, COR_H):
            surf.append(np.stack([X.ravel(), Y.ravel(), np.full(X.size, z)], axis=1))
        X, Z = np.meshgrid(xs, zs, indexing="ij")
        for y in (-COR_HY, COR_HY):
            surf.append(np.stack([X.ravel(), np.full(X.size, y), Z.ravel()], axis=1))
        Y, Z = np.meshgrid(ys, zs, indexing="ij")
        for x in (-COR_HX, COR_HX):
            surf.append(np.stack([np.full(Y.size, x), Y.ravel(), Z.ravel()], axis=1))
        corner = []
        xe = np.arange(-COR_HX, COR_HX + 1e-9, corner_spacing)
        for y in (-COR_HY, COR_HY):
            for z in (0.0, COR_H):
                corner.append(np.stack([xe, np.full(len(xe), y), np.full(len(xe), z)], axis=1))
        zc = np.arange(0.0, COR_H + 1e-9, corner_spacing)
        for x in (-COR_HX, COR_HX):
            for y in (-COR_HY, COR_HY):
                corner.append(np.stack([np.full(len(zc), x), np.full(len(zc), y), zc], axis=1))
        for x0, x1, y0, y1 in self.pilasters:
            y_in = y0 if abs(y0) < abs(y1) else y1
            for x in (x0, x1):
                corner.append(np.stack([np.full(len(zc), x), np.full(len(zc), y_in), zc], axis=1))
            zz = np.arange(sp / 2, COR_H, sp)
            surf.append(np.stack([np.full(len(zz), 0.5 * (x0 + x1)), np.full(len(zz), y_in), zz], axis=1))
        surf, corner = np.concatenate(surf), np.concatenate(corner)
        return corner + rng.normal(0, jitter, corner.shape), surf + rng.normal(0, jitter, surf.shape)
