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
lib = h.lib

d_pts = torch.from_numpy(pts).to(dev)
d_ring = torch.from_numpy(ring.astype(np.int16)).to(dev)
d_full = torch.empty((n, 4), dtype=torch.float32, device=dev)
d_fring = torch.empty(n, dtype=torch.int16, device=dev)
d_curv = torch.empty(n, dtype=torch.float32, device=dev)
d_label = torch.empty(n, dtype=torch.uint8, device=dev)
d_idx = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(4)]
d_cnt = [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(5)]
d_status = torch.empty(B, dtype=torch.int32, device=dev)
d_corner = torch.empty((n, 4), dtype=torch.float32, device=dev)
d_surf = torch.empty((n, 4), dtype=torch.float32, device=dev)
d_map_c = torch.from_numpy(map_c).to(dev)
d_map_s = torch.from_numpy(map_s).to(dev)
d_guess = torch.from_numpy(guess).to(dev)
d_poses = torch.empty_like(d_guess)
d_mstat = torch.zeros(B, dtype=torch.int32, device=dev)
corner_off = np.zeros(B + 1, np.int32)
surf_off = np.zeros(B + 1, np.int32)

f = capi.FeaturesBatch()
f.full_pts, f.full_ring, f.curvature, f.label = d_full.data_ptr(), d_fring.data_ptr(), d_curv.data_ptr(), d_label.data_ptr()
f.sharp_idx, f.less_sharp_idx, f.flat_idx, f.less_flat_idx = (t.data_ptr() for t in d_idx)
f.n_full, f.n_sharp, f.n_less_sharp, f.n_flat, f.n_less_flat = (t.data_ptr() for t in d_cnt)
vp = C.c_void_p


def check(s, what):
    if s != 0:
        raise RuntimeError("%s: status %d %s" % (what, s, lib.msfl_last_error(h.h).decode()))


def extract():
    check(lib.msfl_extract_features_batch(h.h, C.c_int(B), vp(d_pts.data_ptr()), vp(d_ring.data_ptr()), off.ctypes.data_as(vp), C.byref(f),
                                          vp(d_status.data_ptr()), C.c_int(capi.MEM_DEVICE)), "extract")


def voxel():
    # corner (0.2 m) and surf (0.4 m) lists in one call: both filters are enqueued before the one synchronisation
    check(lib.msfl_voxel_downsample_batch_pair(h.h, C.c_int(B), vp(d_full.data_ptr()), off.ctypes.data_as(vp),
                                               vp(d_idx[1].data_ptr()), vp(d_cnt[2].data_ptr()), C.c_float(0.2), vp(d_corner.data_ptr()), corner_off.ctypes.data_as(vp),
                                               vp(d_idx[3].data_ptr()), vp(d_cnt[4].data_ptr()), C.c_float(0.4), vp(d_surf.data_ptr()), surf_off.ctypes.data_as(vp),
                                               C.c_int(capi.MEM_DEVICE)), "voxel corner + surf")


def register():
    d_poses.copy_(d_guess)
    h.set_map(d_map_c, d_map_s, len(map_c), len(map_s), capi.MEM_DEVICE)
    h.match_scan2map_batch_device(B, d_corner, corner_off, d_surf, surf_off, d_poses, d_mstat)


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
poses = d_poses.cpu().numpy()
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
