"""Device-resident batch pipeline: raw VLP-16 scans -> feature extraction -> voxel filter (corner 0.2 m,
surf 0.4 m) -> scan-to-map registration, B scans per call, nothing but offsets and counts visiting
the host.  Reports per-stage and end-to-end throughput and checks the poses against ground truth
and against the host-memory single-scan path.  Run on the GPU box:
    python tools/pipeline_throughput.py [B] [steps] [beams: 16|64] [map_points]
BASELINE.json shapes per GPU: configs[1] = 1024 16 200000; configs[3] (10k x 64-beam over 8 GPUs) = 1250 3 64 200000;
configs[4] (5k scans vs a 2M-point map over 8 GPUs) = 625 5 16 2000000."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from msf_loam_amd import capi, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
BEAMS = int(sys.argv[3]) if len(sys.argv) > 3 else 16
MAP_POINTS = int(sys.argv[4]) if len(sys.argv) > 4 else 200000
SCAN_KW = dict(n_beams=64, n_az=1900, elev=(-24.8, 2.0)) if BEAMS == 64 else {}
dev = torch.device("cuda", 0)
world = synth.World(ground_half=synth.ground_half_for_target(MAP_POINTS))
map_c, map_s = synth.make_map(world)
truth = synth.random_poses(B, synth.SEED + 900)
rng = np.random.default_rng(9)
guess = np.stack([synth.perturb_pose(p, rng, 0.3, 3.0) for p in truth])
scans = [synth.make_scan(world, truth[i], synth.SEED + 901 + i, **SCAN_KW) for i in range(B)]
pts = np.concatenate([p for p, _ in scans])
ring = np.concatenate([r for _, r in scans])
off = np.cumsum([0] + [len(p) for p, _ in scans]).astype(np.int32)
n = len(pts)

torch.zeros(1, device=dev)
h = capi.Handle(0)
h.set_stream(torch.cuda.current_stream(dev).cuda_stream)
from msf_loam_amd.pipeline import BatchPipeline
pipe = BatchPipeline(h, pts, ring, off, dev)
pipe.set_map(map_c, map_s)
d_guess = torch.from_numpy(guess).to(dev)
extract, voxel = pipe.extract, pipe.voxel


def register():
    pipe.register(d_guess)


def timed(fn, k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k


def whole():
    extract(); voxel(); register()


whole()                                                     # warm-up (allocations)
whole()
import gc
gc.collect(); gc.disable()                                   # a gen-2 collection of this harness is a ~40 ms pause
t_ext, t_vox, t_reg = timed(extract, STEPS), timed(voxel, STEPS), timed(register, STEPS)
t_ext2 = timed(extract, STEPS)
t_all = timed(whole, STEPS)
poses = pipe.d_poses.cpu().numpy()
corner_off, surf_off, d_mstat = pipe.corner_off, pipe.surf_off, pipe.d_mstat
err_t = max(synth.pose_error(poses[i], truth[i])[0] for i in range(B))
err_r = max(synth.pose_error(poses[i], truth[i])[1] for i in range(B))

# the same scans through the host-memory single-scan calls: must agree bit for bit
h2 = capi.Handle(0)
h2.set_map(map_c, map_s)
same = True
for i in (0, B // 2, B - 1):
    fe = h2.extract_features(*scans[i])
    c = h2.voxel_downsample(fe["full"][fe["less_sharp"]], 0.2)
    s = h2.voxel_downsample(fe["full"][fe["less_flat"]], 0.4)
    _, p, _ = h2.match_scan2map(c, s, guess[i])
    same = same and np.array_equal(p, poses[i])
print(json.dumps({"pipeline": {"scans": B, "beams": BEAMS, "map_points": int(len(map_c) + len(map_s)), "points": int(n), "features_after_voxel": int(corner_off[-1] + surf_off[-1]),
                               "ms_extract": 1e3 * t_ext, "ms_extract_again": 1e3 * t_ext2, "ms_voxel": 1e3 * t_vox, "ms_register(incl. map index)": 1e3 * t_reg,
                               "ms_end_to_end": 1e3 * t_all, "scans_per_s_end_to_end": B / t_all,
                               "max_pose_error_vs_truth": [err_t, err_r], "equals_host_single_scan_path": bool(same),
                               "status_ok": int((d_mstat.cpu().numpy() == 0).sum())}}))
