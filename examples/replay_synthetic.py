#!/usr/bin/env python3
"""Sequential odometry + mapping replay on a synthetic trajectory (the stand-in for BASELINE
configs[2], `nsh_indoor_outdoor.bag`, which is not available offline).

Per scan, like MSF_LOAM's LaserOdometry::AddLaserScan -> LaserMapping::Run (laser_odometry.cc:69-95,
laser_mapping.cc:138-258), LiDAR-only:
    extract features (stage A) -> MatchScan2Scan against the previous scan (stage B)
    -> pose_odom chain -> voxel down-sample (0.2 / 0.4 m) -> MatchScan2Map against the accumulated
    map (stage C) -> TransformUpdate -> insert the scan's features into the map.
The map store is the device-resident HybridGrid replacement (msfl_grid_*):
InsertScan2Map / GetSurroundedCloud as in laser_mapping.cc:273-278,330-338.  `backend` is either the
GPU library or, in tests, the CPU oracle driven through the same loop, so the two trajectories can
be compared pose by pose.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msf_loam_amd import synth  # noqa: E402


def compose(a, b):
    """Rigid3d operator* (rigid_transform.h:105-111)."""
    Ra = synth.quat_to_matrix(a[3:])
    q = synth.quat_mul(a[3:], b[3:])
    return np.r_[Ra @ b[:3] + a[:3], q / np.linalg.norm(q)]


def inverse(a):
    qc = np.r_[-a[3:6], a[6]]
    return np.r_[-(synth.quat_to_matrix(qc) @ a[:3]), qc]


def transform_cloud(pose, pts):
    out = pts.copy()
    out[:, :3] = (pts[:, :3].astype(np.float64) @ synth.quat_to_matrix(pose[3:]).T + pose[:3]).astype(np.float32)
    return out


def trajectory(n, seed=synth.SEED + 300):
    """Smooth loop inside the room: ~0.25 m and ~1.5 deg per scan."""
    rng = np.random.default_rng(seed)
    poses = []
    for k in range(n):
        a = 2 * np.pi * k / max(n, 120)
        x, y = 9.0 * np.cos(a), 6.0 * np.sin(a)
        yaw = a + np.pi / 2
        poses.append(np.r_[x, y, 1.8 + 0.02 * np.sin(3 * a), synth.quat_from_euler(0.01 * np.sin(2 * a), 0.01 * np.cos(a), yaw)])
    del rng
    return np.array(poses)


class GpuBackend:
    def __init__(self, device=0):
        from msf_loam_amd import capi
        self.odo, self.mapper = capi.Handle(device), capi.Handle(device)    # two handles, like the reference's two matchers
        self.capi = capi

    def extract(self, pts, ring):
        return self.odo.extract_features(pts, ring)

    def voxel(self, pts, leaf):
        return self.mapper.voxel_downsample(pts, leaf)

    def scan2scan(self, last, cur, pose):
        s, p, _ = self.odo.match_scan2scan(last["full"][last["less_sharp"]], last["ring"][last["less_sharp"]],
                                           last["full"][last["less_flat"]], last["ring"][last["less_flat"]],
                                           cur["full"][cur["sharp"]], cur["full"][cur["flat"]], pose)
        return p

    def scan2map(self, mc, ms, corner, surf, pose):
        self.mapper.set_map(mc, ms)
        s, p, _ = self.mapper.match_scan2map(corner, surf, pose)
        return p

    def new_grids(self):
        return self.capi.Grid(self.mapper, 3.0, 0.2), self.capi.Grid(self.mapper, 3.0, 0.4)   # laser_mapping.cc:44-45,60-68


def synthetic_imu(poses_true, switch_at=50, seed=synth.SEED + 900, samples=45):
    """Per-scan IMU inputs for the replay (msfl_slam_imu): a synthetic pre-integration (45 samples over 0.11 s: a small steady
    rotation + drift, so that the de-skew passes change every point by millimetres), the estimator "initialised" from scan
    `switch_at` on (estimator.h:57: 50 scans), the pre-solved pose = the true pose nudged by a few millimetres (an IMU
    prediction that does not depend on earlier results, so the pipelined form can be fed without waiting)."""
    rng = np.random.default_rng(seed)
    out = []
    for k, pose in enumerate(poses_true):
        sum_dt = np.linspace(0.0, 0.11, samples)
        w = rng.normal(0, 0.02, 3)                                   # rad/s
        acc = rng.normal(0, 0.05, 3)
        v_body = rng.normal(0, 0.03, 3)
        ang = sum_dt[:, None] * w[None, :]
        half = 0.5 * np.linalg.norm(ang, axis=1)
        axis = w / max(np.linalg.norm(w), 1e-12)
        dq = np.c_[np.sin(half)[:, None] * axis[None, :], np.cos(half)]
        dp = v_body[None, :] * sum_dt[:, None] + 0.5 * acc[None, :] * sum_dt[:, None] ** 2
        init = k >= switch_at
        pres = pose.copy()
        pres[:3] += rng.normal(0, 0.004, 3)
        dth = rng.normal(0, 0.001, 3)
        q = synth.quat_mul(pose[3:], np.r_[0.5 * dth, 1.0])
        pres[3:] = q / np.linalg.norm(q)
        out.append(dict(sum_dt=sum_dt, delta_q=dq, delta_p=dp, is_initialized=init, velocity=rng.normal(0, 0.05, 3),
                        gravity=np.array([0.0, 0.0, 0.3]) + rng.normal(0, 0.01, 3), presolved_pose=pres))
    return out


def run(backend, world, poses_true, verbose=False, maps_out=None, scans=None, quirks=False, imu=None, clouds_out=None):
    """quirks: FilterLessFlatLessCornerFeature as the reference executes it (laser_mapping.cc:340-364: the surf cloud cut to its first
    n_less_sharp points).  imu: per-scan dicts (synthetic_imu) -> UndistortScan before the match while not initialised
    (:170-176), the is_initialized matcher branch + DoUndistort before the insert afterwards (:197-211); needs a backend with
    undistort / deskew / scan2map_deskew (the oracle backend of the tests)."""
    # a backend may bring its own Rigid3d algebra (the oracle-driven loop of the tests uses the oracle's quaternion forms)
    compose_ = getattr(backend, "compose", compose)
    inverse_ = getattr(backend, "inverse", inverse)
    transform_ = getattr(backend, "transform", transform_cloud)
    n = len(poses_true)
    odo2first = np.array([0, 0, 0, 0, 0, 0, 1.0])        # pose_scan2world_ (odometry frame = first scan)
    curr2last = np.array([0, 0, 0, 0, 0, 0, 1.0])
    odom2map = poses_true[0].copy()                       # anchor the map frame at the true first pose
    grid_c, grid_s = backend.new_grids()                  # hybrid_grid_map_corner_ / hybrid_grid_map_surf_
    last = None
    est, t_stage = [], dict(extract=0.0, odometry=0.0, voxel=0.0, surround=0.0, mapping=0.0, insert=0.0)
    for k in range(n):
        pts, ring = scans[k] if scans is not None else synth.make_scan(world, poses_true[k], synth.SEED + 5000 + k)
        t0 = time.perf_counter(); f = backend.extract(pts, ring); t1 = time.perf_counter()
        if last is not None:
            curr2last = backend.scan2scan(last, f, curr2last)                 # laser_odometry.cc:75 (guess = last delta)
            odo2first = compose_(odo2first, curr2last)                        # :79
        t2 = time.perf_counter()
        # ---- mapping thread (LaserMapping::Run): its own copies of the two clouds that reach the pose and the map
        ls, lf = f["full"][f["less_sharp"]], f["full"][f["less_flat"]]
        im = imu[k] if imu is not None else None
        if im is not None and im["sum_dt"] is None:
            im = None
        pre = (im["sum_dt"], im["delta_q"], im["delta_p"]) if im is not None else None
        if im is not None and not im["is_initialized"]:                        # UndistortScan, laser_mapping.cc:170-176
            ls, lf = backend.undistort(pre, ls), backend.undistort(pre, lf)
        if quirks:                                                            # FilterLessFlatLessCornerFeature, :186,340-364
            assert len(ls) <= len(lf), "the reference reads out of bounds here"
            lf = lf[:len(ls)]
        corner = backend.voxel(ls, 0.2)                                       # laser_mapping.cc:264-270
        surf = backend.voxel(lf, 0.4)
        t3 = time.perf_counter()
        pose_map = compose_(odom2map, odo2first)                              # TransformAssociateToMap, laser_mapping.h:55-57
        map_c = grid_c.get_surrounded(ls, pose_map)                           # GetSurroundedCloud on the UN-down-sampled
        map_s = grid_s.get_surrounded(lf, pose_map)                           # feature clouds, laser_mapping.cc:273-278
        t3b = time.perf_counter()
        if len(map_c) > 10 and len(map_s) > 50:                               # gate, laser_mapping.cc:284-285
            if im is not None and im["is_initialized"]:                       # mapping_scan_matcher.cc:28-59: start from the pre-solve
                pose_map = backend.scan2map_deskew(map_c, map_s, corner, surf, pre, im["velocity"], im["gravity"],
                                                   np.array(im["presolved_pose"], np.float64))
            else:
                pose_map = backend.scan2map(map_c, map_s, corner, surf, pose_map)
        t4 = time.perf_counter()
        odom2map = compose_(pose_map, inverse_(odo2first))                    # TransformUpdate, laser_mapping.h:59-61
        if im is not None and im["is_initialized"]:                           # DoUndistort, laser_mapping.cc:197-211 (pose_odom_scan2world_!)
            ls = backend.deskew(pre, ls, odo2first[3:], im["velocity"], im["gravity"])
            lf = backend.deskew(pre, lf, odo2first[3:], im["velocity"], im["gravity"])
        if clouds_out is not None:
            # the data products that reach neither the pose nor the map: cloud_full_res after the scan's IMU passes (UndistortScan
            # :170-176 / DoUndistort :206) and TransformPointCloud(cloud_full_res, pose_map_scan2world_) (:214-217)
            full = f["full"]
            if im is not None and not im["is_initialized"]:
                full = backend.undistort(pre, full)
            elif im is not None:
                full = backend.deskew(pre, full, odo2first[3:], im["velocity"], im["gravity"])
            clouds_out.append(dict(full_scan=full, full_map=transform_(pose_map, full), ring=f["ring"], sharp=f["sharp"], less_sharp=f["less_sharp"],
                                   flat=f["flat"], less_flat=f["less_flat"]))
        grid_c.insert_scan(transform_(pose_map, ls))                          # InsertScan2Map, laser_mapping.cc:330-338
        grid_s.insert_scan(transform_(pose_map, lf))
        t5 = time.perf_counter()
        last = f
        est.append(pose_map)
        for name, dt in (("extract", t1 - t0), ("odometry", t2 - t1), ("voxel", t3 - t2), ("surround", t3b - t3),
                         ("mapping", t4 - t3b), ("insert", t5 - t4)):
            if k >= 2:
                t_stage[name] += dt
        if verbose and k % 20 == 0:
            print(k, synth.pose_error(pose_map, poses_true[k]), grid_c.size(), grid_s.size(), file=sys.stderr)
    m = max(n - 2, 1)
    if maps_out is not None:
        maps_out["corner"], maps_out["surf"] = grid_c.dump(), grid_s.dump()
    return np.array(est), {k: 1e3 * v / m for k, v in t_stage.items()}


def world_drive(world, kind, n):
    """A drive through the outdoor / corridor world (~0.4 m and a degree or two per scan): down the x = 0 street following the relief, or
    down the corridor's axis (the same drives as tests/common.py:world_drive; wraps around after ~125 scans so that any length stays inside)."""
    poses = []
    for k in range(n):
        if kind == "outdoor":
            kk = k % 250
            kk = kk if kk < 125 else 250 - kk                        # there and back again
            y = -25.0 + 0.4 * kk
            x = 1.2 * np.sin(0.11 * kk)
            yaw = np.pi / 2 - np.arctan(1.2 * 0.11 * np.cos(0.11 * kk) / 0.4) * 0.5
            z = float(world.geom.ground(x, y)) + 1.8 + 0.02 * np.sin(0.5 * kk)
            poses.append(np.r_[x, y, z, synth.quat_from_euler(0.015 * np.sin(0.3 * kk), 0.01 * np.cos(0.2 * kk), yaw)])
        else:
            kk = k % 250
            kk = kk if kk < 125 else 250 - kk
            poses.append(np.r_[-20.0 + 0.4 * kk, 0.3 * np.sin(0.15 * kk), 1.5 + 0.02 * np.sin(0.4 * kk),
                               synth.quat_from_euler(0.01 * np.sin(0.3 * kk), 0.01 * np.cos(0.25 * kk), 0.05 * np.sin(0.2 * kk))])
    return np.array(poses)


def run_slam(world, poses_true, pipelined=False, device=0, scans=None, verbose=False, maps_out=None, quirks=False, imu=None, clouds_out=None):
    """The same loop through the device-resident SLAM step (msfl_slam_add_scan): raw scan in, pose out, one
    synchronisation per scan (pipelined=False) or none until the record is fetched one scan later (pipelined=True: the
    odometry chain of scan k + 1 runs under the mapping chain of scan k, like the reference's two threads).
    Returns (poses, records, wall-clock ms per scan over the scans after the second)."""
    from msf_loam_amd import capi
    n = len(poses_true)
    if scans is None:
        scans = [synth.make_scan(world, poses_true[k], synth.SEED + 5000 + k) for k in range(n)]
    cap = max(len(p) for p, _ in scans)
    rings = int(max(r.max() for _, r in scans)) + 1
    if os.environ.get("MSFL_REPLAY_DEFAULT_CAPS") == "1":          # msfl_slam_default_config's launch bounds (200 000 points, 128 rings) instead of tight ones
        cap, rings = 200000, 128
    slam = capi.Slam(device, max_scan_points=cap, max_rings=rings, pose_odom2map=poses_true[0],
                     reference_quirks=1 if quirks else 0, keep_clouds=1 if clouds_out is not None else 0)
    recs = [None] * n
    t_start = None
    for k in range(n):
        if k == 2:
            t_start = time.perf_counter()
        im = imu[k] if imu is not None else None
        if pipelined:
            slam.add_scan(*scans[k], wait=False, imu=im)
            if k >= 1:
                recs[k - 1] = slam.result(k - 1)
                if clouds_out is not None:
                    clouds_out.append(slam.clouds(k - 1))
        else:
            recs[k] = slam.add_scan(*scans[k], imu=im)
            if clouds_out is not None:
                clouds_out.append(slam.clouds(k))
        if verbose and k % 50 == 0 and recs[max(k - 1, 0)] is not None:
            r = recs[max(k - 1, 0)]
            print(k, list(r.grid_corner)[:3], list(r.grid_surf)[:3], r.status_mapping, file=sys.stderr)
    if pipelined:
        recs[n - 1] = slam.result(n - 1)
        if clouds_out is not None:
            clouds_out.append(slam.clouds(n - 1))
    wall = time.perf_counter() - t_start if t_start is not None else 0.0
    est = np.array([np.array(r.pose_map[:]) for r in recs])
    if maps_out is not None:
        gc_, gs_ = slam.grids()
        maps_out["corner"], maps_out["surf"] = gc_.dump(), gs_.dump()
    slam.close()
    return est, recs, 1e3 * wall / max(n - 2, 1)


def ate(est, truth):
    return float(np.sqrt(np.mean(np.sum((est[:, :3] - truth[:, :3]) ** 2, axis=1))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=300)
    ap.add_argument("--mode", choices=["slam", "slam-pipelined", "staged"], default="slam",
                    help="slam: msfl_slam_add_scan, one synchronisation per scan; slam-pipelined: results fetched one scan late; "
                         "staged: the round-2 loop of single-stage host-pointer calls")
    ap.add_argument("--dump-poses", default=None, help="write the estimated map poses (n x 7 float64, .npy) here")
    ap.add_argument("--reference-quirks", action="store_true", help="msfl_slam_config.reference_quirks = 1 (surf list truncation of the reference)")
    ap.add_argument("--imu", action="store_true", help="feed a synthetic pre-integration with every scan (UndistortScan, then the is_initialized branch from scan 50)")
    ap.add_argument("--world", choices=["room", "outdoor", "corridor"], default="room",
                    help="room: the SURVEY 8d world and its loop; outdoor / corridor: msf_loam_amd/worlds.py with a drive down the street / the axis (round 5)")
    ap.add_argument("--beams", type=int, choices=[16, 64], default=16, help="64: the KITTI-scale sensor of BASELINE configs[3] (64 x 1 900, -24.8 .. +2 degrees)")
    ap.add_argument("--keep-clouds", action="store_true", help="msfl_slam_config.keep_clouds = 1 and fetch every scan's clouds (what PublishScan would publish)")
    args = ap.parse_args()
    if args.world == "room":
        world = synth.World(ground_half=45.0)
        truth = trajectory(args.scans)
    else:
        world = synth.World(kind=args.world)
        truth = world_drive(world, args.world, args.scans)
    if args.mode == "staged":
        est, ms = run(GpuBackend(0), world, truth, verbose=True)
        print(json.dumps({"mode": "staged", "scans": args.scans, "ate_rmse_m": ate(est, truth),
                          "final_error_m_rad": synth.pose_error(est[-1], truth[-1]),
                          "latency_ms_per_scan": ms, "note": "single-scan host-pointer calls (includes PCIe staging)"}))
        return
    kw = dict(n_beams=64, n_az=1900, elev=(-24.8, 2.0)) if args.beams == 64 else {}
    scans = [synth.make_scan(world, truth[k], synth.SEED + 5000 + k, **kw) for k in range(args.scans)]
    import gc
    gc.collect(); gc.disable()
    est, recs, ms = run_slam(world, truth, pipelined=args.mode == "slam-pipelined", scans=scans, quirks=args.reference_quirks,
                             imu=synthetic_imu(truth) if args.imu else None, clouds_out=[] if args.keep_clouds else None)
    last = recs[-1]
    if args.dump_poses:
        np.save(args.dump_poses, est)
    print(json.dumps({"mode": args.mode, "world": args.world, "beams": args.beams, "keep_clouds": bool(args.keep_clouds), "scans": args.scans, "reference_quirks": bool(args.reference_quirks), "imu": bool(args.imu),
                      "ate_rmse_m": ate(est, truth),
                      "final_error_m_rad": synth.pose_error(est[-1], truth[-1]), "ms_per_scan_end_to_end": ms,
                      "scans_per_s": 1e3 / ms if ms else None,
                      "map_points": [last.grid_corner[0], last.grid_surf[0]], "map_cells": [last.grid_corner[1], last.grid_surf[1]],
                      "mapping_gate_closed_scans": int(sum(1 for r in recs if r.status_mapping != 0)),
                      "mean_features_after_voxel": [float(np.mean([r.n_corner_ds for r in recs])), float(np.mean([r.n_surf_ds for r in recs]))],
                      "mean_surrounded_map_points": [float(np.mean([r.n_map_corner for r in recs])), float(np.mean([r.n_map_surf for r in recs]))],
                      "note": "raw host scan in (18 B/pt over PCIe), 480-byte record out; wall clock around the calls"}))


if __name__ == "__main__":
    main()
