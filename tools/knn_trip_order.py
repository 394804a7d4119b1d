#!/usr/bin/env python3
"""Upper bound of "order each scan's 5-NN queries by predicted trip count" (VERDICT r02, next-round item 3), measured
WITHOUT building the pre-pass: the prediction is computed here with torch (map points in the 27 one-metre cells around the
query transformed by its initial guess — what a device pre-pass would read off cell_start differences), every scan's
corner and surf features are permuted by it on the host, and the unchanged product kernel is timed on the permuted batch
(HIP events inside the library, msfl_set_timing).  A registration does not depend on the order of its features except
through the rounding of the solver's sums, so the poses stay valid.  Orders compared:
  original      ring / azimuth order out of the extraction + voxel filter
  by_count      predicted candidate count, descending (lanes of a wavefront get similar trip counts)
  by_count_cell by the query's map cell first (locality), then count
  shuffled      a random permutation per scan (control: no locality, no homogeneity)
Prints one JSON line; run on the GPU box:  python tools/knn_trip_order.py [steps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from msf_loam_amd import capi, synth

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
h = capi.Handle(0)
B = 1024
inp = bench.build_inputs(B, 200000, 0, bench.product_extractor(h))
stream = torch.cuda.current_stream(dev)
h.set_stream(stream.cuda_stream)
co, so = inp["corner_off"], inp["surf_off"]


def predicted_counts(feat, off, map_pts):
    """map points in the 3 x 3 x 3 block of 1 m cells around every feature, transformed by its scan's guess"""
    m = torch.from_numpy(map_pts[:, :3]).to(dev)
    lo = m.min(0).values - 2.0
    K = int((m.max(0).values - lo).max().item()) + 4

    def key(c):
        return (c[:, 2] * K + c[:, 1]) * K + c[:, 0]
    mc = torch.floor(m - lo).long()
    hist = torch.bincount(key(mc), minlength=K * K * K)
    f = torch.from_numpy(feat[:, :3].astype(np.float64)).to(dev)
    scan_of = torch.from_numpy(np.repeat(np.arange(B), np.diff(off))).to(dev)
    g = torch.from_numpy(inp["guesses"]).to(dev)
    q = g[scan_of, 3:]                                              # [x y z w]
    t = g[scan_of, :3]
    qv = q[:, :3]
    uv = 2.0 * torch.cross(qv, f, dim=1)
    w = f + q[:, 3:4] * uv + torch.cross(qv, uv, dim=1) + t
    c = torch.floor(w.float() - lo).long().clamp(1, K - 2)
    cnt = torch.zeros(len(f), dtype=torch.long, device=dev)
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                cnt += hist[key(c + torch.tensor([dx, dy, dz], device=dev))]
    return cnt.cpu().numpy(), key(c).cpu().numpy()


cnt_c, cell_c = predicted_counts(inp["corner"], co, inp["map_corner"])
cnt_s, cell_s = predicted_counts(inp["surf"], so, inp["map_surf"])
rng = np.random.default_rng(5)


def permuted(kind):
    out = {}
    for name, feat, off, cnt, cell in (("corner", inp["corner"], co, cnt_c, cell_c), ("surf", inp["surf"], so, cnt_s, cell_s)):
        perm = np.arange(len(feat))
        for b in range(B):
            lo, hi = off[b], off[b + 1]
            if kind == "by_count":
                order = np.argsort(-cnt[lo:hi], kind="stable")
            elif kind == "by_count_cell":
                order = np.lexsort((-cnt[lo:hi], cell[lo:hi]))
            elif kind == "shuffled":
                order = rng.permutation(hi - lo)
            else:
                order = np.arange(hi - lo)
            perm[lo:hi] = lo + order
        out[name] = feat[perm]
    return out


d_map_c = torch.from_numpy(inp["map_corner"]).to(dev)
d_map_s = torch.from_numpy(inp["map_surf"]).to(dev)
d_guess = torch.from_numpy(inp["guesses"]).to(dev)
d_poses = torch.empty_like(d_guess)
d_status = torch.zeros(B, dtype=torch.int32, device=dev)
res = {}
ref_poses = None
for kind in ("original", "by_count", "by_count_cell", "shuffled", "original"):
    f = permuted(kind)
    d_corner, d_surf = torch.from_numpy(f["corner"]).to(dev), torch.from_numpy(f["surf"]).to(dev)

    def step():
        d_poses.copy_(d_guess)
        h.set_map(d_map_c, d_map_s, len(inp["map_corner"]), len(inp["map_surf"]), capi.MEM_DEVICE)
        h.match_scan2map_batch_device(B, d_corner, co, d_surf, so, d_poses, d_status)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    h.set_timing(1); h.get_timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(STEPS):
        step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / STEPS
    t = h.get_timing(reset=True)
    h.set_timing(3); h.get_timing(reset=True); step(); torch.cuda.synchronize()
    cand = h.get_timing(reset=True).knn_candidates
    h.set_timing(0)
    poses = d_poses.cpu().numpy()
    if ref_poses is None:
        ref_poses = poses
    dmax = max(synth.pose_error(poses[i], ref_poses[i])[0] for i in range(B))
    key = kind if kind not in res else kind + "_again"
    res[key] = {"assoc_ms": t.ms_assoc / max(t.launches_assoc, 1), "fit_ms": t.ms_fit / max(t.launches_fit, 1),
                "solve_ms": t.ms_solve / max(t.launches_solve, 1), "step_ms_with_event_timers": 1e3 * wall,
                "candidates_per_query": cand / 2 / float(co[-1] + so[-1]), "max_pose_shift_vs_original_m": dmax,
                "failed": int((d_status.cpu().numpy() != 0).sum())}
print(json.dumps({"knn_trip_order": res, "features": int(co[-1] + so[-1]),
                  "predicted_count_mean": float((cnt_c.sum() + cnt_s.sum()) / (len(cnt_c) + len(cnt_s)))}))
