"""Round 5 additions to bench.py's `stages` (never part of `value`; measured after everything that feeds it):

  worlds.{outdoor,corridor}   the registration step of bench.py (map index rebuilt + B scans registered, device resident) in
                              the harness's other two worlds (msf_loam_amd/worlds.py): registrations/s, per-kernel ms, 5-NN
                              candidates per query, failed scans, a 3-scan oracle spot-check
  config3_share               BASELINE configs[3] per GPU: 1 250 64-beam scans (25 distinct sweeps x 50 guesses) through the
                              device-resident pipeline, per stage
  config4_share               BASELINE configs[4] per GPU: 625 scans against a 2 M-point map
  pairs                       `msfl_match_pairs_batch`: 256 (map, scan) pairs with 256 different ~34 k-point maps

Checker use only: the oracle module is handed in by bench.py for the spot-checks; this file never imports it.
Stand-alone on the GPU box (timings only):  python tools/bench_worlds.py [worlds|shares|all] [scans] [outdoor,corridor]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _timed(fn, k, sync):
    sync()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    sync()
    return (time.perf_counter() - t0) / k


def _register_stats(h, pipe, d_guess, sync, reps):
    """Times pipe.register (index build + registration, device resident) and splits it by kernel class with the library's
    event timers; one more step with the counting 5-NN instantiation."""
    pipe.register(d_guess); pipe.register(d_guess)
    t_reg = _timed(lambda: pipe.register(d_guess), reps, sync)
    h.set_timing(1); h.get_timing(reset=True)
    for _ in range(3):
        pipe.register(d_guess)
    sync()
    t = h.get_timing(reset=True)
    k_ms = {"assoc": t.ms_assoc / max(t.launches_assoc, 1), "fit": t.ms_fit / max(t.launches_fit, 1),
            "solve": t.ms_solve / max(t.launches_solve, 1), "index_build": t.ms_index / max(t.launches_index, 1)}
    h.set_timing(3); h.get_timing(reset=True)
    pipe.register(d_guess); sync()
    tc = h.get_timing(reset=True)
    h.set_timing(0)
    n_feat = int(pipe.corner_off[-1] + pipe.surf_off[-1])
    cand = (tc.knn_candidates + tc.knn_candidates_seeded) / 2.0 / max(n_feat, 1)
    return t_reg, k_ms, cand, n_feat


def _slam_replay(kind, n, beams=16):
    """BASELINE configs[2]'s per-scan step on a drive through that world, in torch-free child processes like tools/bench_stages.py
    (examples/replay_synthetic.py --world): ms per scan synchronous / pipelined, ATE, gate-closed scans."""
    import subprocess
    out = {}
    for mode in ("slam", "slam-pipelined"):
        cp = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "replay_synthetic.py"), "--scans", str(n), "--mode", mode, "--world", kind, "--beams", str(beams)],
                            capture_output=True, text=True, timeout=600)
        if cp.returncode != 0:
            raise RuntimeError("SLAM replay child failed: " + cp.stderr[-2000:])
        out[mode] = json.loads([l for l in cp.stdout.splitlines() if l.startswith("{")][-1])
    a, b = out["slam"], out["slam-pipelined"]
    return {"scans": n, "beams": beams, "ms_per_scan_synchronous": a["ms_per_scan_end_to_end"], "ms_per_scan_pipelined": b["ms_per_scan_end_to_end"],
            "ate_rmse_m": a["ate_rmse_m"], "same_final_error": a["final_error_m_rad"] == b["final_error_m_rad"],
            "map_points": a["map_points"], "mean_surrounded_map_points": a["mean_surrounded_map_points"],
            "mean_features_after_voxel": a["mean_features_after_voxel"], "mapping_gate_closed_scans": a["mapping_gate_closed_scans"],
            "note": ("64 x 1 900 returns per sweep (the sensor of BASELINE configs[3]); parity of the 64-beam step against the oracle-driven loop: "
                     "tests/test_dataset_io.py::test_kitti_layout_replays_through_the_device_resident_slam_step[64]; a short replay: the first scans' allocations weigh in")
                    if beams == 64 else
                    ("parity of this replay against the oracle-driven loop: tests/test_gpu_replay.py::test_device_resident_slam_step_in_the_other_worlds; "
                     "the corridor's ATE is LOAM's own failure along the unobservable axis (the CPU loop's too)")}


def measure_worlds(device=0, checker=None, scans=256, reps=5, spot=3, kinds=("outdoor", "corridor"), copies=4, slam_scans=60):
    """`scans` distinct sweeps per world, each registered from `copies` different guesses: scans x copies registrations per step (1 024
    by default, the batch of bench.py's headline, so that registrations/s compare like for like; registrations are independent units)."""
    import torch
    from msf_loam_amd import capi, synth
    from msf_loam_amd.pipeline import BatchPipeline
    dev = torch.device("cuda", device)
    sync = lambda: torch.cuda.synchronize(dev)   # noqa: E731
    out = {"note": "bench.py's step (map index rebuilt + all scans registered, device resident) in the other synthetic worlds; "
                   "features from the product's own extraction + voxel kernels; mean of %d steps" % reps}
    for kind in kinds:
        t0 = time.perf_counter()
        w = synth.World(kind=kind)
        mc, ms = synth.make_map(w)
        truth1 = synth.world_poses(w, scans, synth.SEED + 2)
        raw1 = [synth.make_scan(w, truth1[i], synth.SEED + 100 + i) for i in range(scans)]
        order = np.tile(np.arange(scans), copies)
        n_sweeps, scans = scans, scans * copies
        truth = truth1[order]
        rng = np.random.default_rng(synth.SEED + 3)
        guess = np.stack([synth.perturb_pose(p, rng) for p in truth])
        raw = [raw1[i] for i in order]
        t_prep = time.perf_counter() - t0
        pts = np.concatenate([p for p, _ in raw]); ring = np.concatenate([r for _, r in raw])
        off = np.cumsum([0] + [len(p) for p, _ in raw]).astype(np.int32)
        h = capi.Handle(device)
        h.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        pipe = BatchPipeline(h, pts, ring, off, dev)
        pipe.set_map(mc, ms)
        d_guess = torch.from_numpy(guess).to(dev)
        pipe.extract(); pipe.voxel()
        t_ext = _timed(pipe.extract, reps, sync)
        t_vox = _timed(pipe.voxel, reps, sync)
        t_reg, k_ms, cand, n_feat = _register_stats(h, pipe, d_guess, sync, reps)
        poses = pipe.d_poses.cpu().numpy(); status = pipe.d_mstat.cpu().numpy()
        err = np.array([synth.pose_error(poses[b], truth[b]) for b in range(scans)])
        gerr = np.array([synth.pose_error(guess[b], truth[b]) for b in range(scans)])
        chk = None
        if checker is not None:
            dc, ds = pipe.d_corner.cpu().numpy(), pipe.d_surf.cpu().numpy()
            dmax, same_counts = 0.0, True
            for b in np.linspace(0, scans - 1, spot).astype(int):
                c = dc[pipe.corner_off[b]:pipe.corner_off[b + 1]]; s = ds[pipe.surf_off[b]:pipe.surf_off[b + 1]]
                rc, po, _ = checker.match_scan2map(mc, ms, c, s, guess[b])
                dmax = max(dmax, *synth.pose_error(poses[b], po))
                same_counts = same_counts and rc == int(status[b])
            chk = {"scans": spot, "max_pose_delta": dmax, "tolerance": 1e-4, "ok": bool(dmax < 1e-4 and same_counts)}
        out[kind] = {"scans": scans, "distinct_sweeps": n_sweeps, "map_points": int(len(mc) + len(ms)), "map_corner": int(len(mc)), "map_surf": int(len(ms)),
                     "points_per_scan": float(off[-1]) / scans, "features_per_scan": n_feat / scans,
                     "ms_per_step": 1e3 * t_reg, "registrations_per_s": scans / t_reg,
                     "kernels_ms": k_ms, "knn": {"candidates_per_query": cand},
                     "ms_extract": 1e3 * t_ext, "ms_voxel": 1e3 * t_vox,
                     "n_failed": int((status != 0).sum()),
                     "median_pose_error_vs_truth_m_rad": [float(np.median(err[:, 0])), float(np.median(err[:, 1]))],
                     "median_guess_error_m_rad": [float(np.median(gerr[:, 0])), float(np.median(gerr[:, 1]))],
                     "oracle_spot_check": chk, "prep_s": t_prep,
                     "reference": "mapping_scan_matcher.cc:19-278 on worlds.py:" + kind}
        del pipe
        h.close()
        scans = n_sweeps
        if slam_scans > 0:
            out[kind]["slam_step"] = _slam_replay(kind, slam_scans)
            if kind == "outdoor":                       # the KITTI-scale sensor (64 x 1 900 returns, ~100 k-point less-flat lists) where it belongs
                out[kind]["slam_step_64_beams"] = _slam_replay(kind, max(20, slam_scans // 2), beams=64)
    return out


def measure_shares(device=0, checker=None, reps=3, spot=3, which=("config3_share", "config4_share", "pairs")):
    import torch
    from msf_loam_amd import capi, synth
    from msf_loam_amd.pipeline import BatchPipeline
    dev = torch.device("cuda", device)
    sync = lambda: torch.cuda.synchronize(dev)   # noqa: E731
    out = {}

    def share(tag, sweeps, copies, mc, ms, note):
        B = len(sweeps) * copies
        order = np.repeat(np.arange(len(sweeps)), copies)
        truth = np.stack([sweeps[i][2] for i in order])
        rng = np.random.default_rng(77)
        guess = np.stack([synth.perturb_pose(p, rng, 0.3, 3.0) for p in truth])
        pts = np.concatenate([sweeps[i][0] for i in order]); ring = np.concatenate([sweeps[i][1] for i in order])
        off = np.cumsum([0] + [len(sweeps[i][0]) for i in order]).astype(np.int32)
        h = capi.Handle(device)
        h.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        pipe = BatchPipeline(h, pts, ring, off, dev)
        pipe.set_map(mc, ms)
        d_guess = torch.from_numpy(guess).to(dev)
        pipe.run(d_guess); pipe.run(d_guess)
        t_ext = _timed(pipe.extract, reps, sync)
        t_vox = _timed(pipe.voxel, reps, sync)
        t_reg, k_ms, cand, n_feat = _register_stats(h, pipe, d_guess, sync, reps)
        t_all = _timed(lambda: pipe.run(d_guess), reps, sync)
        poses = pipe.d_poses.cpu().numpy(); status = pipe.d_mstat.cpu().numpy()
        err = np.array([synth.pose_error(poses[b], truth[b]) for b in range(B)])
        chk = None
        if checker is not None:
            dmax = 0.0
            for b in np.linspace(0, B - 1, spot).astype(int):
                c = pipe.d_corner[pipe.corner_off[b]:pipe.corner_off[b + 1]].cpu().numpy()
                s = pipe.d_surf[pipe.surf_off[b]:pipe.surf_off[b + 1]].cpu().numpy()
                rc, po, _ = checker.match_scan2map(mc, ms, c, s, guess[b])
                dmax = max(dmax, *synth.pose_error(poses[b], po))
            chk = {"scans": spot, "max_pose_delta": dmax, "tolerance": 1e-4, "ok": bool(dmax < 1e-4)}
        out[tag] = {"scans": B, "distinct_sweeps": len(sweeps), "points": int(off[-1]), "map_points": int(len(mc) + len(ms)),
                    "features_after_voxel": n_feat, "ms_extract": 1e3 * t_ext, "ms_voxel": 1e3 * t_vox, "ms_register_incl_index": 1e3 * t_reg,
                    "ms_end_to_end": 1e3 * t_all, "scans_per_s_end_to_end": B / t_all, "kernels_ms": k_ms,
                    "knn": {"candidates_per_query": cand}, "n_failed": int((status != 0).sum()),
                    "max_pose_error_vs_truth_m_rad": [float(err[:, 0].max()), float(err[:, 1].max())],
                    "oracle_spot_check": chk, "note": note}
        del pipe
        h.close()

    if "config3_share" in which:
        w = synth.World(ground_half=synth.ground_half_for_target(200000))
        mc, ms = synth.make_map(w)
        poses = synth.random_poses(25, synth.SEED + 640)
        sweeps = [synth.make_scan(w, poses[i], synth.SEED + 641 + i, n_beams=64, n_az=1900, elev=(-24.8, 2.0)) + (poses[i],) for i in range(25)]
        share("config3_share", sweeps, 50, mc, ms,
              "BASELINE configs[3] per GPU: 10 000 64-beam scans over 8 GPUs = 1 250 per GPU; 25 distinct sweeps, each from 50 guesses "
              "(tests/test_gpu_scale_configs.py does the same); raw scans -> extraction -> voxel filters -> registration, device resident")
    if "config4_share" in which:
        w = synth.World(ground_half=synth.ground_half_for_target(2_000_000))
        mc, ms = synth.make_map(w)
        poses = synth.random_poses(125, synth.SEED + 2500)
        sweeps = [synth.make_scan(w, poses[i], synth.SEED + 2501 + i) + (poses[i],) for i in range(125)]
        share("config4_share", sweeps, 5, mc, ms,
              "BASELINE configs[4] per GPU: 5 000 registrations against a 2 M-point map over 8 GPUs = 625 per GPU; 125 distinct sweeps x 5 guesses")
    if "pairs" in which:
        P, target = 256, 30000
        rng = np.random.default_rng(12)
        mcs, mss, cs, ss, guesses, truths = [], [], [], [], [], []
        for p in range(P):
            w = synth.World(seed=synth.SEED + 700 + p, ground_half=synth.ground_half_for_target(target))
            mc, ms = synth.make_map(w, seed=synth.SEED + 1700 + p)
            truth = synth.random_poses(1, synth.SEED + 2700 + p)[0]
            pts, ring, kind = synth.make_scan(w, truth, synth.SEED + 3700 + p, with_kind=True)
            c, s = synth.direct_features(pts, kind)
            mcs.append(mc); mss.append(ms); cs.append(c); ss.append(s); guesses.append(synth.perturb_pose(truth, rng)); truths.append(truth)
        cat = lambda ls: (np.concatenate(ls), np.cumsum([0] + [len(a) for a in ls]).astype(np.int32))   # noqa: E731
        (mc, mco), (ms, mso), (c, co), (s, so) = cat(mcs), cat(mss), cat(cs), cat(ss)
        guesses = np.array(guesses)
        h = capi.Handle(device)
        h.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        d = {k: torch.from_numpy(v).to(dev) for k, v in dict(mc=mc, ms=ms, c=c, s=s, g=guesses).items()}
        d_poses = torch.empty_like(d["g"]); d_status = torch.zeros(P, dtype=torch.int32, device=dev)

        def run():
            d_poses.copy_(d["g"])
            h.match_pairs_batch_device(P, d["mc"], mco, d["ms"], mso, d["c"], co, d["s"], so, d_poses, d_status)
        for _ in range(3):
            run()
        dt = _timed(run, max(reps, 5), sync)
        h.set_timing(1); h.get_timing(reset=True)
        run(); sync()
        t = h.get_timing(reset=True); h.set_timing(0)
        poses = d_poses.cpu().numpy()
        err = np.array([synth.pose_error(poses[p], truths[p]) for p in range(P)])
        chk = None
        if checker is not None:
            dmax = 0.0
            for p in np.linspace(0, P - 1, spot).astype(int):
                rc, po, _ = checker.match_scan2map(mcs[p], mss[p], cs[p], ss[p], guesses[p])
                dmax = max(dmax, *synth.pose_error(poses[p], po))
            chk = {"pairs": spot, "max_pose_delta": dmax, "tolerance": 1e-4, "ok": bool(dmax < 1e-4)}
        out["pairs"] = {"pairs": P, "map_points_total": int(len(mc) + len(ms)), "map_points_per_pair": int((len(mc) + len(ms)) / P),
                        "features_total": int(len(c) + len(s)), "ms_per_call": 1e3 * dt, "pairs_per_s": P / dt,
                        "kernels_ms": {"index_build_both_kinds": t.ms_index, "knn": t.ms_assoc, "fit": t.ms_fit, "solve": t.ms_solve},
                        "max_pose_error_vs_truth_m_rad": [float(err[:, 0].max()), float(err[:, 1].max())],
                        "n_failed": int((d_status.cpu().numpy() != 0).sum()), "oracle_spot_check": chk,
                        "note": "north star's 'many map-submap pairs': one msfl_match_pairs_batch call, P different maps, device resident",
                        "reference": "one MatchScan2Map per pair, laser_mapping.cc:304-311"}
        h.close()
    return out


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    res = {}
    if what in ("worlds", "all"):
        kinds = tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 else ("outdoor", "corridor")
        res["worlds"] = measure_worlds(scans=n, kinds=kinds)
    if what in ("shares", "all"):
        res.update(measure_shares())
    print(json.dumps(res))
