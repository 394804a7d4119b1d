"""CPU model A/B of the 5-NN grid's cell edge ("density-adaptive cell edge: 0.5 x gate with reach 2 stays exact").

For the room and the outdoor world: the surf features of real scans (oracle extraction + 0.4 m voxel filter, in the voxel order the
kernel sees them, transformed by the perturbed guess) searched in the world's surf map with
    A  cell = 1.001 m, reach 1 (3 x 3 rows of 7 x-sub-cells)        the shipped grid
    B  cell = 0.5005 m, reach 2 (5 x 5 rows of 13 x-sub-cells)      the proposed finer grid (exact for the same 1 m gate)
Rows visited by increasing lower bound, skipped on the bound, end cells trimmed against the running 5th distance (the shipped walk's
rules).  Reported per query: candidates, rows visited, insertions; per wavefront of 64 consecutive queries: the trip count the SIMD
pays = sum over the visit positions of the longest lane's candidate count / 2 (two candidates per iteration), and the visit
positions that any lane uses (each costs the whole wavefront two dependent loads + the loop prologue).
    python tools/sim/knn_cell_ab.py [room|outdoor] [scans] [queries per scan]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from msf_loam_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (harness tool: features as the matcher receives them)


class Grid:
    def __init__(self, M, cell, reach, xs=3):
        self.cell, self.reach, self.xs = float(cell), reach, xs
        self.o = M.min(0)
        edge = np.array([cell / xs, cell, cell])
        self.dims = (np.floor((M.max(0) - self.o) / edge) + 2).astype(int)
        c = np.floor((M - self.o) / edge).astype(int)
        key = (c[:, 2] * self.dims[1] + c[:, 1]) * self.dims[0] + c[:, 0]
        order = np.argsort(key, kind="stable")
        self.S = M[order]
        self.start = np.searchsorted(key[order], np.arange(int(np.prod(self.dims)) + 1))

    def walk(self, q, gate=1.0):
        cell, xs, R = self.cell, self.xs, self.reach
        u = (q - self.o) / np.array([cell / xs, cell, cell])
        qc = np.floor(u).astype(int)
        gap = lambda v, c: max(max(c - v, v - (c + 1)) - 1e-3, 0.0)   # noqa: E731
        rows = []
        for dy in range(-R, R + 1):
            for dz in range(-R, R + 1):
                y, z = qc[1] + dy, qc[2] + dz
                if 0 <= y < self.dims[1] and 0 <= z < self.dims[2]:
                    rows.append(((gap(u[1], y) ** 2 + gap(u[2], z) ** 2) * cell * cell, y, z))
        rows.sort()
        best = []
        per_pos, ncand, nins = [], 0, 0
        x_lo, x_hi = max(qc[0] - R * xs, 0), min(qc[0] + R * xs, self.dims[0] - 1)
        for row2, y, z in rows:
            d4 = best[4] if len(best) >= 5 else gate
            if row2 > d4:
                continue
            room = d4 - row2
            a, b = x_lo, x_hi
            sub2 = (cell / xs) ** 2
            while a < qc[0] and gap(u[0], a) ** 2 * sub2 > room:
                a += 1
            while b > qc[0] and gap(u[0], b) ** 2 * sub2 > room:
                b -= 1
            base = (z * self.dims[1] + y) * self.dims[0]
            s, e = self.start[base + a], self.start[base + b + 1]
            per_pos.append(e - s)
            if e > s:
                d = ((self.S[s:e] - q) ** 2).sum(1)
                for dd in d:
                    ncand += 1
                    if dd <= (best[4] if len(best) >= 5 else gate):
                        nins += 1
                        best.append(float(dd)); best.sort(); del best[5:]
        return ncand, nins, per_pos


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "outdoor"
    n_scans = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    per_scan = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    orc.build()
    w = synth.World(kind=kind) if kind != "room" else synth.World(ground_half=synth.ground_half_for_target(200000))
    _, ms = synth.make_map(w)
    M = ms[:, :3].astype(np.float64)
    grids = {"A cell 1.001 reach 1": Grid(M, 1.001, 1), "B cell 0.5005 reach 2": Grid(M, 0.5005, 2)}
    poses = synth.world_poses(w, n_scans, synth.SEED + 2)
    rng = np.random.default_rng(synth.SEED + 3)
    out = {k: dict(q=0, cand=0, ins=0, rows=0, waves=0, trips=0, positions=0, cand_hist=[]) for k in grids}
    for i, T in enumerate(poses):
        pts, ring = synth.make_scan(w, T, synth.SEED + 100 + i)
        f = orc.extract_features(pts, ring)
        surf = orc.voxel_grid(f["full"][f["less_flat"]], 0.4)[:, :3].astype(np.float64)
        guess = synth.perturb_pose(T, rng)
        Q = surf @ synth.quat_to_matrix(guess[3:]).T + guess[:3]
        start = int(rng.integers(0, max(len(Q) - per_scan, 1)))
        Q = Q[start:start + per_scan]                     # consecutive features: whole wavefronts
        for name, g in grids.items():
            o = out[name]
            res = [g.walk(q) for q in Q]
            for c, n_, pp in res:
                o["q"] += 1; o["cand"] += c; o["ins"] += n_; o["rows"] += len(pp); o["cand_hist"].append(c)
            for wv in range(0, len(res) - 63, 64):
                lanes = [r[2] for r in res[wv:wv + 64]]
                npos = max(len(p) for p in lanes)
                trips = sum((max((p[k] if k < len(p) else 0) for p in lanes) + 1) // 2 for k in range(npos))
                o["waves"] += 1; o["trips"] += trips; o["positions"] += npos
    print("world %s: %d surf-map points, %d queries" % (kind, len(M), out[next(iter(out))]["q"]))
    for name, o in out.items():
        h = np.array(o["cand_hist"])
        print("%-24s candidates/query %.1f (median %d, p90 %d, p99 %d, max %d)  insertions %.1f  rows visited %.2f | per wavefront: pair-iterations %.1f, "
              "visit positions %.1f" % (name, o["cand"] / o["q"], np.median(h), np.percentile(h, 90), np.percentile(h, 99), h.max(), o["ins"] / o["q"],
                                        o["rows"] / o["q"], o["trips"] / max(o["waves"], 1), o["positions"] / max(o["waves"], 1)))


if __name__ == "__main__":
    main()
