"""The stages of the hot path that `bench.py`'s headline does not time (the raw-scan -> pose
pipeline, BASELINE configs[2]'s per-scan SLAM step), measured after the timed loop of bench.py on the same box and put into its
JSON line as `stages` (never part of `value`).  Every stage carries a 3-scan spot-check against a CHECKER object handed in by the
caller (bench.py passes the CPU oracle it already uses for its cpu_baseline / pose-delta leg; this file never imports it) and names
the rocprofv3 kernel rows that back it (`profiles/r04_*`).

Also runnable on its own on the GPU box (no checker: timings only):  python tools/bench_stages.py [scans] [slam_scans]
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))

HBM_PEAK_GBS = 8000.0


def _timed(fn, k, sync):
    sync()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    sync()
    return (time.perf_counter() - t0) / k


def measure(raw, world, map_corner, map_surf, truth, guesses, device=0, reps=5, slam_scans=100, spot=3, checker=None):
    """raw: [(pts, ring, kind)] of B scans seen from `truth` (bench.py's own batch); returns the `stages` dict.
    checker: the CPU restatement (bench.py's oracle module) for the spot-checks; None = timings only."""
    import torch
    from msf_loam_amd import capi, synth
    from msf_loam_amd.pipeline import BatchPipeline
    import replay_synthetic as rp
    orc = checker
    if orc is None:
        spot = 0

    dev = torch.device("cuda", device)
    sync = lambda: torch.cuda.synchronize(dev)   # noqa: E731
    B = len(raw)
    pts = np.concatenate([p for p, _, _ in raw])
    ring = np.concatenate([r for _, r, _ in raw])
    off = np.cumsum([0] + [len(p) for p, _, _ in raw]).astype(np.int32)
    n = int(off[-1])
    h = capi.Handle(device)
    h.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    out = {"note": "untimed for `value`; wall clock around device-resident calls (torch.cuda.synchronize on both sides), mean of %d" % reps}

    # ---------------- stage A: feature extraction (msf_loam_node.cc:160-378), B scans per call, device resident
    pipe = BatchPipeline(h, pts, ring, off, dev)
    pipe.set_map(map_corner, map_surf)
    pipe.extract(); pipe.extract()
    t_ext = _timed(pipe.extract, reps, sync)
    h.set_timing(1); h.get_timing(reset=True)
    pipe.extract(); sync()
    ev_ext = h.get_timing(reset=True).ms_extract
    h.set_timing(0)
    cnt = [c.cpu().numpy() for c in pipe.d_cnt]
    full = pipe.d_full.cpu().numpy(); idx = [t.cpu().numpy() for t in pipe.d_idx]
    ok = True
    for b in np.linspace(0, B - 1, spot).astype(int):
        fo = orc.extract_features(raw[b][0], raw[b][1])
        o = off[b]
        ok = ok and np.array_equal(full[o:o + cnt[0][b]], fo["full"])
        for li, key in enumerate(("sharp", "less_sharp", "flat", "less_flat")):
            ok = ok and np.array_equal(idx[li][o:o + cnt[li + 1][b]], fo[key])
    alg = 32.0 * n
    out["extract"] = {"scans": B, "points": n, "ms": 1e3 * t_ext, "ms_kernels_hip_events": ev_ext, "scans_per_s": B / t_ext,
                      "algorithmic_bytes": alg, "GBps": alg / t_ext / 1e9, "frac": alg / t_ext / 1e9 / HBM_PEAK_GBS,
                      "oracle_spot_check": {"scans": spot, "bit_exact_lists_and_full_cloud": bool(ok)} if orc else None,
                      "kernels": ["extract_prepare_kernel", "extract_curvature_kernel", "extract_pick_kernel", "extract_compact_kernel"],
                      "reference": "msf_loam_node.cc:85-350 (A1-A9)"}

    # ---------------- the two voxel filters (laser_mapping.cc:264-270)
    pipe.voxel(); pipe.voxel()
    t_vox = _timed(pipe.voxel, reps, sync)
    dc, ds = pipe.d_corner.cpu().numpy(), pipe.d_surf.cpu().numpy()
    okv = True
    for b in np.linspace(0, B - 1, spot).astype(int):
        o = off[b]
        f_full = full[o:o + cnt[0][b]]
        oc = orc.voxel_grid(f_full[idx[1][o:o + cnt[2][b]]], 0.2)
        os_ = orc.voxel_grid(f_full[idx[3][o:o + cnt[4][b]]], 0.4)
        okv = okv and np.array_equal(dc[pipe.corner_off[b]:pipe.corner_off[b + 1]], oc) and np.array_equal(ds[pipe.surf_off[b]:pipe.surf_off[b + 1]], os_)
    n_in = int(cnt[2].sum() + cnt[4].sum())
    out["voxel"] = {"clouds": 2 * B, "points_in": n_in, "points_out": int(pipe.corner_off[-1] + pipe.surf_off[-1]), "ms": 1e3 * t_vox,
                    "algorithmic_bytes": 16.0 * (n_in + int(pipe.corner_off[-1] + pipe.surf_off[-1])),
                    "GBps": 16.0 * (n_in + int(pipe.corner_off[-1] + pipe.surf_off[-1])) / t_vox / 1e9,
                    "oracle_spot_check": {"scans": spot, "bit_exact": bool(okv)} if orc else None,
                    "kernels": ["voxel_cloud_lds_kernel"], "reference": "laser_mapping.cc:264-270 (pcl::VoxelGrid x2)"}

    # ---------------- raw scans -> poses, batched (stage A -> voxel -> stage C with the map index rebuilt)
    d_guess = torch.from_numpy(np.ascontiguousarray(guesses)).to(dev)
    whole = lambda: pipe.run(d_guess)   # noqa: E731
    whole(); whole()
    t_all = _timed(whole, reps, sync)
    poses = pipe.d_poses.cpu().numpy()
    okp, dmax = True, 0.0
    for b in np.linspace(0, B - 1, spot).astype(int):
        rc, po, _ = orc.match_scan2map(map_corner, map_surf, dc[pipe.corner_off[b]:pipe.corner_off[b + 1]], ds[pipe.surf_off[b]:pipe.surf_off[b + 1]], guesses[b])
        dt, dr = synth.pose_error(poses[b], po)
        dmax = max(dmax, dt, dr)
    err = [synth.pose_error(poses[b], truth[b]) for b in range(B)]
    out["pipeline"] = {"scans": B, "ms": 1e3 * t_all, "scans_per_s": B / t_all, "max_pose_error_vs_truth_m_rad": [max(e[0] for e in err), max(e[1] for e in err)],
                       "oracle_spot_check": {"scans": spot, "max_pose_delta": dmax, "tolerance": 1e-4, "ok": bool(dmax < 1e-4)} if orc else None,
                       "stages": "msfl_extract_features_batch -> msfl_voxel_downsample_batch_pair -> msfl_set_map + msfl_match_scan2map_batch",
                       "reference": "msf_loam_node.cc:160-378 -> laser_mapping.cc:264-270 -> mapping_scan_matcher.cc:19-278"}

    # ---------------- stage B: scan-to-scan (odometry_scan_matcher.cc:43-285), B pairs (scan b seen again after a small motion)
    rng = np.random.default_rng(9)
    nxt_pose = [synth.perturb_pose(truth[b], rng, 0.25, 2.0) for b in range(B)]
    nxt = [synth.make_scan(world, nxt_pose[b], synth.SEED + 5901 + b) for b in range(B)]
    offb = np.cumsum([0] + [len(p) for p, _ in nxt]).astype(np.int32)
    fa = h.extract_features_batch(pts, ring, off)
    fb = h.extract_features_batch(np.concatenate([p for p, _ in nxt]), np.concatenate([r for _, r in nxt]), offb)

    def cat(fs, key, with_ring=False):
        p = np.concatenate([f["full"][f[key]] for f in fs]); o = np.cumsum([0] + [len(f[key]) for f in fs]).astype(np.int32)
        return p, (np.concatenate([f["ring"][f[key]] for f in fs]) if with_ring else None), o
    sets = [cat(fa, "less_sharp", True), cat(fa, "less_flat", True), cat(fb, "sharp"), cat(fb, "flat")]
    ident = np.tile([0, 0, 0, 0, 0, 0, 1.0], (B, 1))
    dsets, keep = [], []
    for pk, rk, ok_ in sets:
        tp = torch.from_numpy(np.ascontiguousarray(pk, np.float32)).to(dev)
        tr = torch.from_numpy(np.ascontiguousarray(rk if rk is not None else np.zeros(len(pk)), np.uint16).view(np.int16)).to(dev)
        to = np.ascontiguousarray(ok_, np.int32)
        rb = capi.RingCloudBatch(); rb.pts, rb.ring, rb.off = tp.data_ptr(), tr.data_ptr(), to.ctypes.data
        dsets.append(rb); keep.append((tp, tr, to))
    d_ident = torch.from_numpy(ident).to(dev)
    d_pose = torch.empty_like(d_ident)
    d_st = torch.zeros(B, dtype=torch.int32, device=dev)

    def run_b():
        d_pose.copy_(d_ident)
        s = h.lib.msfl_match_scan2scan_batch(h.h, C.c_int(B), C.byref(dsets[0]), C.byref(dsets[1]), C.byref(dsets[2]), C.byref(dsets[3]),
                                             C.c_void_p(d_pose.data_ptr()), C.c_void_p(d_st.data_ptr()), None, C.c_int(capi.MEM_DEVICE))
        assert s == 0, s
    run_b(); run_b()
    t_b = _timed(run_b, reps, sync)
    pb, stb = d_pose.cpu().numpy(), d_st.cpu().numpy()
    dmaxb = 0.0
    for b in np.linspace(0, B - 1, spot).astype(int):
        a, c = fa[b], fb[b]
        rc, po, _ = orc.match_scan2scan(a["full"][a["less_sharp"]], a["ring"][a["less_sharp"]], a["full"][a["less_flat"]], a["ring"][a["less_flat"]],
                                        c["full"][c["sharp"]], c["full"][c["flat"]], ident[0])
        dmaxb = max(dmaxb, *synth.pose_error(pb[b], po))
    out["scan2scan"] = {"pairs": B, "ok": int((stb == 0).sum()), "ms": 1e3 * t_b, "pairs_per_s": B / t_b,
                        "queries": int(sets[2][2][-1] + sets[3][2][-1]), "targets": int(sets[0][2][-1] + sets[1][2][-1]),
                        "oracle_spot_check": {"pairs": spot, "max_pose_delta": dmaxb, "tolerance": 1e-4, "ok": bool(dmaxb < 1e-4)} if orc else None,
                        "kernels": ["odom_bin_kernel", "assoc_scan2scan_grid_kernel", "assoc_scan2scan_kernel", "lm_solve_kernel"],
                        "reference": "odometry_scan_matcher.cc:43-285 (B1-B3)"}
    del keep
    h.close()

    # ---------------- BASELINE configs[2]: the per-scan SLAM step on a synthetic drive (the real bag is not in the image).
    # Run in CHILD processes that do not import torch: the step's wall clock depends on the HIP runtime the process loaded, and
    # a process that imported torch first runs on torch's bundled runtime (measured: 0.44 -> 0.55-0.71 ms per scan pipelined,
    # tools/slam_probe.py); a C++ host of the library (the reference is one) loads the system runtime like these children do.
    if slam_scans > 0:
        import subprocess
        import tempfile
        sw = synth.World(ground_half=45.0)
        tr_ = rp.trajectory(slam_scans)
        scans = [synth.make_scan(sw, tr_[k], synth.SEED + 5000 + k) for k in range(min(slam_scans, 8))]
        runs = {}
        with tempfile.TemporaryDirectory() as td:
            for mode in ("slam", "slam-pipelined"):
                dump = os.path.join(td, mode + ".npy")
                cp = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "replay_synthetic.py"), "--scans", str(slam_scans), "--mode", mode,
                                     "--dump-poses", dump], capture_output=True, text=True, timeout=600)
                if cp.returncode != 0:
                    raise RuntimeError("SLAM replay child failed: " + cp.stderr[-2000:])
                runs[mode] = (json.loads([l for l in cp.stdout.splitlines() if l.startswith("{")][-1]), np.load(dump))
        (js, est_s), (jp, est_p) = runs["slam"], runs["slam-pipelined"]
        ms_s, ms_p = js["ms_per_scan_end_to_end"], jp["ms_per_scan_end_to_end"]

        class _O:      # the oracle-driven loop of tests/test_gpu_replay.py on the first scans
            def __init__(self): self.o = orc
            def extract(self, p, r): return orc.extract_features(p, r)
            def voxel(self, p, leaf): return orc.voxel_grid(p, leaf)
            def scan2scan(self, last, cur, pose):
                return orc.match_scan2scan(last["full"][last["less_sharp"]], last["ring"][last["less_sharp"]], last["full"][last["less_flat"]],
                                           last["ring"][last["less_flat"]], cur["full"][cur["sharp"]], cur["full"][cur["flat"]], pose)[1]
            def scan2map(self, mc, ms, c, s, pose): return orc.match_scan2map(mc, ms, c, s, pose)[1]
            def new_grids(self): return orc.HybridGrid(3.0, 0.2), orc.HybridGrid(3.0, 0.4)
            def compose(self, a, b): return orc.pose_compose(a, b)
            def inverse(self, a):
                qc = np.r_[-np.asarray(a[3:6]), a[6]]
                return np.r_[-orc.quat_rotate(qc, np.asarray(a[:3], np.float64)), qc]
            def transform(self, pose, p): return orc.transform_cloud(p, pose)
        n_chk = min(max(spot, 8), slam_scans) if orc else 0
        d = 0.0
        if n_chk:
            est_o, _ = rp.run(_O(), sw, tr_[:n_chk], scans=scans[:n_chk])
            d = max(max(synth.pose_error(a, b)) for a, b in zip(est_s[:n_chk], est_o))
        out["slam_step"] = {"scans": slam_scans, "points_per_scan": int(np.mean([len(p) for p, _ in scans])),
                            "ms_per_scan_synchronous": ms_s, "ms_per_scan_pipelined": ms_p, "scans_per_s_pipelined": 1e3 / ms_p if ms_p else None,
                            "ate_rmse_m": rp.ate(est_s, tr_), "pipelined_equals_synchronous_bitwise": bool(np.array_equal(est_s, est_p)),
                            "mapping_gate_closed_scans": js["mapping_gate_closed_scans"],
                            "oracle_spot_check": {"scans": n_chk, "max_pose_delta_vs_oracle_loop": d, "tolerance": 1e-6, "ok": bool(d < 1e-6)} if orc else None,
                            "note": "raw host scan in (18 B/pt over PCIe), result record out; wall clock around msfl_slam_add_scan in child "
                                    "processes without torch (examples/replay_synthetic.py); synthetic substitute for nsh_indoor_outdoor.bag "
                                    "(absent from the image)",
                            "reference": "msf_loam_node.cc:160-378 -> laser_odometry.cc:69-95 -> laser_mapping.cc:138-338"}
    return out


if __name__ == "__main__":
    from msf_loam_amd import synth
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    slam_n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    world = synth.World(ground_half=synth.ground_half_for_target(200000))
    mc, ms = synth.make_map(world)
    truth = synth.random_poses(B, synth.SEED + 2)
    rng = np.random.default_rng(synth.SEED + 3)
    guesses = np.stack([synth.perturb_pose(p, rng) for p in truth])
    raw = [synth.make_scan(world, truth[i], synth.SEED + 100 + i, with_kind=True) for i in range(B)]
    print(json.dumps(measure(raw, world, mc, ms, truth, guesses, slam_scans=slam_n)))
