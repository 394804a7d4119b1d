#!/usr/bin/env python3
"""Map-store cost against map size: msfl_grid_insert_scan / msfl_grid_get_surrounded timed while the map grows.

Each inserted "scan" is 6 000 points on a patch of ground and walls that moves 2 m per scan (a vehicle's down-sampled
less-flat list lands in ~150 cells of 3 m); the map therefore grows by a few thousand voxels per scan.  Prints one JSON
line per checkpoint: map points / cells, ms per insert and ms per surround query (wall clock around the C call, host
buffers, so the figures include the PCIe copy of the scan and, for the query, of the result).
"""
import argparse, gc, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from msf_loam_amd import capi


def scan_at(rng, k, n):
    c = np.array([2.0 * k, 0.3 * np.sin(0.05 * k) * 40.0, 0.0])
    p = np.empty((n, 4), np.float32)
    m = n // 2
    p[:m, 0] = rng.uniform(-40, 40, m); p[:m, 1] = rng.uniform(-40, 40, m); p[:m, 2] = rng.normal(0, 0.02, m)       # ground
    p[m:, 0] = rng.uniform(-40, 40, n - m); p[m:, 1] = rng.choice([-12.0, 12.0], n - m) + rng.normal(0, 0.02, n - m)
    p[m:, 2] = rng.uniform(0, 6, n - m)                                                                               # walls
    p[:, :3] += c
    p[:, 3] = 0
    return p, c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=1200)
    ap.add_argument("--points", type=int, default=6000)
    ap.add_argument("--every", type=int, default=100)
    a = ap.parse_args()
    rng = np.random.default_rng(7)
    h = capi.Handle()
    g = capi.Grid(h, 3.0, 0.2)
    gc.collect(); gc.disable()
    t_ins, t_sur = [], []
    for k in range(a.scans):
        p, c = scan_at(rng, k, a.points)
        t0 = time.perf_counter(); g.insert_scan(p); t1 = time.perf_counter()
        pose = np.array([c[0], c[1], c[2], 0, 0, 0, 1], np.float64)
        q = p.copy(); q[:, :3] -= c
        t2 = time.perf_counter(); out = g.get_surrounded(q, pose); t3 = time.perf_counter()
        t_ins.append(t1 - t0); t_sur.append(t3 - t2)
        if (k + 1) % a.every == 0:
            n, cells = g.size()
            print(json.dumps({"scans": k + 1, "map_points": n, "cells": cells, "surround_points": len(out),
                              "insert_ms": round(1e3 * float(np.median(t_ins[-a.every:])), 4),
                              "surround_ms": round(1e3 * float(np.median(t_sur[-a.every:])), 4)}), flush=True)
    g.close(); h.close()


if __name__ == "__main__":
    main()
