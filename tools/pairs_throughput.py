"""Throughput of msfl_match_pairs_batch: P (map, scan) pairs with P different ~30 k-point maps, device resident, against the
loop it replaces (msfl_set_map + one msfl_match_scan2map per pair on the same handle).  GPU box:
    python tools/pairs_throughput.py [P] [map_points] [reps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from msf_loam_amd import capi, synth

P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
TARGET = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
rng = np.random.default_rng(12)
mcs, mss, cs, ss, guesses, truths = [], [], [], [], [], []
for p in range(P):
    w = synth.World(seed=synth.SEED + 700 + p, ground_half=synth.ground_half_for_target(TARGET))
    mc, ms = synth.make_map(w, seed=synth.SEED + 1700 + p)
    truth = synth.random_poses(1, synth.SEED + 2700 + p)[0]
    pts, ring, kind = synth.make_scan(w, truth, synth.SEED + 3700 + p, with_kind=True)
    c, s = synth.direct_features(pts, kind)
    mcs.append(mc); mss.append(ms); cs.append(c); ss.append(s); guesses.append(synth.perturb_pose(truth, rng)); truths.append(truth)
cat = lambda ls: (np.concatenate(ls), np.cumsum([0] + [len(a) for a in ls]).astype(np.int32))   # noqa: E731
(mc, mco), (ms, mso), (c, co), (s, so) = cat(mcs), cat(mss), cat(cs), cat(ss)
guesses = np.array(guesses)
h = capi.Handle(0)
h.set_stream(torch.cuda.current_stream(dev).cuda_stream)
d = {k: torch.from_numpy(v).to(dev) for k, v in dict(mc=mc, ms=ms, c=c, s=s, g=guesses).items()}
d_poses = torch.empty_like(d["g"]); d_status = torch.zeros(P, dtype=torch.int32, device=dev)


def run():
    d_poses.copy_(d["g"])
    h.match_pairs_batch_device(P, d["mc"], mco, d["ms"], mso, d["c"], co, d["s"], so, d_poses, d_status)


for _ in range(3):
    run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(REPS):
    run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / REPS
poses = d_poses.cpu().numpy()
err = np.array([synth.pose_error(poses[p], truths[p]) for p in range(P)])
h.set_timing(1); h.get_timing(reset=True)
run(); torch.cuda.synchronize()
t = h.get_timing(reset=True); h.set_timing(0)
# the loop it replaces: one index build + one latency-form registration per pair, device-resident clouds, same handle
d_p1 = torch.empty(7, dtype=torch.float64, device=dev); d_s1 = torch.zeros(1, dtype=torch.int32, device=dev)


def loop():
    for p in range(P):
        h.set_map(d["mc"][mco[p]:mco[p + 1]], d["ms"][mso[p]:mso[p + 1]], int(mco[p + 1] - mco[p]), int(mso[p + 1] - mso[p]), capi.MEM_DEVICE)
        d_p1.copy_(d["g"][p])
        h.match_scan2map_batch_device(1, d["c"][co[p]:co[p + 1]], np.array([0, co[p + 1] - co[p]], np.int32), d["s"][so[p]:so[p + 1]],
                                      np.array([0, so[p + 1] - so[p]], np.int32), d_p1, d_s1)


loop()
torch.cuda.synchronize(); t0 = time.perf_counter()
loop()
torch.cuda.synchronize(); dt_loop = time.perf_counter() - t0
print(json.dumps({"pairs": P, "map_points_total": int(len(mc) + len(ms)), "map_points_per_pair": int((len(mc) + len(ms)) / P),
                  "features_total": int(len(c) + len(s)), "ms_per_call": 1e3 * dt, "pairs_per_s": P / dt,
                  "kernels_ms": {"index_build_both_kinds": t.ms_index, "knn": t.ms_assoc, "fit": t.ms_fit, "solve": t.ms_solve},
                  "max_pose_error_vs_truth_m_rad": [float(err[:, 0].max()), float(err[:, 1].max())], "status_ok": int((d_status.cpu().numpy() == 0).sum()),
                  "loop_of_single_calls": {"ms_total": 1e3 * dt_loop, "pairs_per_s": P / dt_loop, "ms_per_pair": 1e3 * dt_loop / P},
                  "speedup_vs_loop": dt_loop / dt}))
