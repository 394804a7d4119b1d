#!/usr/bin/env python3
"""BASELINE configs[2] on recorded data: replay the PointCloud2 scans of a rosbag (e.g. nsh_indoor_outdoor.bag, reference
README.md:42-47) through the device-resident SLAM step and write the poses as the reference's pose log (proto/msg.proto).
No ROS installation needed (msf_loam_amd/rosbag_io.py).  LiDAR-only: IMU pre-integration and the estimator stay on the
maintainer's side; the IMU samples are copied into the pose log like LaserMapping::AddImu does.

    python examples/replay_bag.py nsh_indoor_outdoor.bag --topic /velodyne_points --imu-topic /imu/data --out poses.pb
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msf_loam_amd import capi, dataset, rosbag_io  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("bag")
    ap.add_argument("--topic", default="/velodyne_points")
    ap.add_argument("--imu-topic", default="/imu/data")
    ap.add_argument("--out", default=None, help="pose log (PbData, proto/msg.proto) to write")
    ap.add_argument("--max-scans", type=int, default=0)
    ap.add_argument("--max-points", type=int, default=200000)
    ap.add_argument("--rings", type=int, default=128)
    ap.add_argument("--reference-quirks", action="store_true")
    args = ap.parse_args()
    slam = capi.Slam(0, max_scan_points=args.max_points, max_rings=args.rings, reference_quirks=1 if args.reference_quirks else 0)
    log = dataset.PoseLog()
    stamps, n, t0 = [], 0, time.perf_counter()
    for topic, mtype, t, data in rosbag_io.BagReader(args.bag).messages(topics=[args.topic, args.imu_topic]):
        if topic == args.imu_topic:
            m = rosbag_io.parse_imu(data)
            log.add_imu(dataset.from_seconds(m["stamp"]), m["linear_acceleration"], m["angular_velocity"])     # laser_mapping.cc:413-418
            continue
        pc = rosbag_io.parse_pointcloud2(data)
        pts, ring = rosbag_io.cloud_to_msfl(pc)
        slam.add_scan(pts, ring, wait=False)                        # two scans in flight, like the reference's two threads
        stamps.append(pc["stamp"])
        if n >= 1:
            r = slam.result(n - 1)
            log.add_odom(dataset.from_seconds(stamps[n - 1]), np.array(r.pose_map[:]))                          # :251-254
        n += 1
        if args.max_scans and n >= args.max_scans:
            break
    if n:
        r = slam.result(n - 1)
        log.add_odom(dataset.from_seconds(stamps[n - 1]), np.array(r.pose_map[:]))
    wall = time.perf_counter() - t0
    if args.out:
        log.save(args.out)
    print(json.dumps({"scans": n, "wall_s": wall, "ms_per_scan_incl_bag_parsing": 1e3 * wall / max(n, 1), "pose_log": args.out}))
    slam.close()


if __name__ == "__main__":
    main()
