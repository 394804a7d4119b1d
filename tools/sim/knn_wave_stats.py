"""Wave-level statistics of the shipped walk: per row position, lanes active and pair-iterations (max over the 64 lanes)."""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))); 
import importlib.util
src = open(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), 'knn_walk_model.py')).read().split("for T in truth:")[0]
exec(src)
def walk_rows(q):
    ux, uy, uz = (q[0] - o[0]) * invx, (q[1] - o[1]) * inv, (q[2] - o[2]) * inv
    qx, qy, qz = int(np.floor(ux)), int(np.floor(uy)), int(np.floor(uz))
    xs, xe = max(qx - XS, 0), min(qx + XS, dims[0] - 1)
    best = []
    cell2 = float(cell) ** 2; cellx2 = (float(cell) / XS) ** 2
    gy = {d: gap(uy, qy + d) for d in (-1, 0, 1)}; gz = {d: gap(uz, qz + d) for d in (-1, 0, 1)}
    sy = -1 if gy[-1] <= gy[1] else 1; sz = -1 if gz[-1] <= gz[1] else 1
    rows = [(0, 0)]
    a = [(sy, 0), (0, sz)]; a.sort(key=lambda r: gy[r[0]] if r[0] else gz[r[1]]); rows += a
    b = [(-sy, 0), (0, -sz)]; b.sort(key=lambda r: gy[r[0]] if r[0] else gz[r[1]]); rows += b
    rows += [(sy, sz)]
    c = [(sy, -sz), (-sy, sz)]; c.sort(key=lambda r: gy[r[0]] ** 2 + gz[r[1]] ** 2); rows += c
    rows += [(-sy, -sz)]
    lens = [0] * 9; ins = [0] * 9
    d4 = lambda: best[4] if len(best) >= 5 else 1.0
    for i, (dy, dz) in enumerate(rows):
        y, z = qy + dy, qz + dz
        if not (0 <= y < dims[1] and 0 <= z < dims[2]): continue
        row2 = (gy[dy] ** 2 + gz[dz] ** 2) * cell2
        if row2 > d4(): continue
        row = (z * dims[1] + y) * dims[0]
        a_, b_ = xs, xe
        room = d4() - row2
        while a_ <= b_ and a_ < qx and gap(ux, a_) ** 2 * cellx2 > room: a_ += 1
        while b_ >= a_ and b_ > qx and gap(ux, b_) ** 2 * cellx2 > room: b_ -= 1
        if a_ > b_: continue
        s, e = start[row + a_], start[row + b_ + 1]
        lens[i] = e - s
        for p in S[s:e]:
            d = float(np.float32((p[0] - q[0]) ** 2 + (p[1] - q[1]) ** 2 + (p[2] - q[2]) ** 2))
            if d <= d4():
                ins[i] += 1; best.append(d); best.sort(); del best[5:]
    return lens, ins
stats = np.zeros((9, 4)); nw = 0; tot_pairs_lane = 0
for T in truth[:2]:
    pts, ring = synth.make_scan(world, T, synth.SEED + 6)
    sub = synth.voxel_downsample_np(pts, 0.4)[:, :3]          # voxel order, like the filter's output
    guess = synth.perturb_pose(T, rng)
    R = synth.quat_to_matrix(guess[3:]); qs = (sub @ R.T + guess[:3]).astype(np.float32)
    for w0 in range(0, min(len(qs), 64 * 12), 64):
        L = np.array([walk_rows(q)[0] for q in qs[w0:w0 + 64]])
        if len(L) < 64: break
        pairs = (L + 1) // 2
        for r in range(9):
            stats[r, 0] += (L[:, r] > 0).mean(); stats[r, 1] += pairs[:, r].max(); stats[r, 2] += pairs[:, r].mean(); stats[r, 3] += (pairs[:, r].max() > 0)
        tot_pairs_lane += pairs.sum(1).max(); nw += 1
print('waves', nw)
print('row  active_frac  wave_pair_iters(max)  mean_pair_iters  visited')
for r in range(9): print(r, np.round(stats[r] / nw, 2))
print('sum of per-row max:', stats[:, 1].sum() / nw, ' max over lanes of per-lane total:', tot_pairs_lane / nw, ' mean lane total:', stats[:, 2].sum() / nw)
