"""CPU model of the 5-NN walk (surf features): candidates per query for (a) the shipped visit, (b) centre row inner cells first."""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from msf_loam_amd import synth
world = synth.World(ground_half=synth.ground_half_for_target(200000))
mc, ms = synth.make_map(world)
M = ms[:, :3].astype(np.float32)
cell = np.float32(1.001); XS = 3
o = M.min(0)
inv = np.float32(1.0) / cell; invx = np.float32(XS) / cell
dims = (np.floor((M.max(0) - o) / np.array([cell / XS, cell, cell])) + 2).astype(int)
cx = np.floor((M[:, 0] - o[0]) * invx).astype(int); cy = np.floor((M[:, 1] - o[1]) * inv).astype(int); cz = np.floor((M[:, 2] - o[2]) * inv).astype(int)
key = (cz * dims[1] + cy) * dims[0] + cx
order = np.argsort(key, kind='stable'); S = M[order]; ks = key[order]
ncell = int(np.prod(dims)); start = np.searchsorted(ks, np.arange(ncell + 1))
rng = np.random.default_rng(3)
truth = synth.random_poses(4, synth.SEED + 5)
tot = {'ship': 0, 'inner': 0, 'ins_ship': 0, 'ins_inner': 0, 'q': 0, 'ranges_ship': 0, 'ranges_inner': 0}
def gap(u, c):
    return max(max(c - u, u - (c + 1)) - 1e-3, 0.0)
def walk(q, mode):
    ux, uy, uz = (q[0] - o[0]) * invx, (q[1] - o[1]) * inv, (q[2] - o[2]) * inv
    qx, qy, qz = int(np.floor(ux)), int(np.floor(uy)), int(np.floor(uz))
    xs, xe = max(qx - XS, 0), min(qx + XS, dims[0] - 1)
    best = []   # sorted distances (top5)
    ncand = nins = nranges = 0
    cell2 = float(cell) ** 2; cellx2 = (float(cell) / XS) ** 2
    gy = {d: gap(uy, qy + d) for d in (-1, 0, 1)}; gz = {d: gap(uz, qz + d) for d in (-1, 0, 1)}
    sy = -1 if gy[-1] <= gy[1] else 1; sz = -1 if gz[-1] <= gz[1] else 1
    rows = [(0, 0)]
    a = [(sy, 0), (0, sz)]; a.sort(key=lambda r: gy[r[0]] if r[0] else gz[r[1]]); rows += a
    b = [(-sy, 0), (0, -sz)]; b.sort(key=lambda r: gy[r[0]] if r[0] else gz[r[1]]); rows += b
    rows += [(sy, sz)]
    c = [(sy, -sz), (-sy, sz)]; c.sort(key=lambda r: gy[r[0]] ** 2 + gz[r[1]] ** 2); rows += c
    rows += [(-sy, -sz)]
    def d4():
        return best[4] if len(best) >= 5 else 1.0
    def scan(row, a_, b_):
        nonlocal ncand, nins, nranges
        if a_ > b_: return
        nranges += 1
        s, e = start[row + a_], start[row + b_ + 1]
        for p in S[s:e]:
            ncand += 1
            d = float(np.float32((p[0] - q[0]) ** 2 + (p[1] - q[1]) ** 2 + (p[2] - q[2]) ** 2))
            if d <= d4():
                nins += 1
                best.append(d); best.sort(); del best[5:]
    def trim(row2, lo, hi):
        a_, b_ = lo, hi
        room = d4() - row2
        while a_ <= b_ and a_ < qx and gap(ux, a_) ** 2 * cellx2 > room: a_ += 1
        while b_ >= a_ and b_ > qx and gap(ux, b_) ** 2 * cellx2 > room: b_ -= 1
        return a_, b_
    for i, (dy, dz) in enumerate(rows):
        y, z = qy + dy, qz + dz
        if not (0 <= y < dims[1] and 0 <= z < dims[2]): continue
        row2 = (gy[dy] ** 2 + gz[dz] ** 2) * cell2
        if row2 > d4(): continue
        row = (z * dims[1] + y) * dims[0]
        if mode == 'inner' and i == 0:
            scan(row, max(xs, qx - 1), min(xe, qx + 1))
            a_, b_ = trim(row2, xs, xe)
            scan(row, a_, min(b_, qx - 2)); scan(row, max(a_, qx + 2), b_)
        else:
            a_, b_ = trim(row2, xs, xe)
            scan(row, a_, b_)
    return ncand, nins, nranges
for T in truth:
    pts, ring = synth.make_scan(world, T, synth.SEED + 6)
    f = synth.direct_features(pts, None) if False else None
    # surf features: voxel-downsample the scan at 0.4 m (a stand-in for less-flat features), transform by a perturbed pose
    sub = synth.voxel_downsample_np(pts, 0.4)[:, :3]; sub = sub[rng.permutation(len(sub))[:500]]
    for guess in (synth.perturb_pose(T, rng), T):
        R = synth.quat_to_matrix(guess[3:]); qs = (sub @ R.T + guess[:3]).astype(np.float32)
        for q in qs:
            for mode in ('ship', 'inner'):
                c, n, r = walk(q, mode)
                tot[mode] += c; tot['ins_' + mode] += n; tot['ranges_' + mode] += r
            tot['q'] += 1
print({k: (v / tot['q'] if k != 'q' else v) for k, v in tot.items()})
