#!/usr/bin/env python3
"""Per-stream timeline of the SLAM step from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace -d gpurun_out/tl -o t -- python examples/replay_synthetic.py --scans 120 --mode slam-pipelined
    python tools/slam_timeline.py gpurun_out/tl/*/t_kernel_trace.csv > profiles/rNN_slam_timeline.md

Every scan starts with one `extract_prepare_kernel` (odometry stream) and ends with one `slam_result_kernel` (mapping
stream).  For the steady state (scans `--skip` .. end) the tool reports, per stream: kernels per scan, busy time (sum of
kernel durations), span (first start to last end of the scan's kernels on that stream) and therefore the idle share of the
span (launch gaps and waits on other streams); and, per kernel name, calls and time per scan.  Streams are named by what
runs on them.
"""
import argparse
import collections
import csv
import statistics
import sys


def short(name):
    n = name.replace("void ", "").replace("msfl::", "").replace("(anonymous namespace)::", "")
    if "rocprim" in n:
        for key in ("radix_sort_block_sort", "merge_sort_block_merge", "merge_sort_block_sort", "radix_sort_onesweep", "onesweep_histograms", "scan_config",
                    "lookback_scan_state", "transform_config", "radix_sort_single", "radix_sort_merge"):
            if key in n:
                return "rocprim:" + key
        return "rocprim:other"
    cut = n.find("(")
    return n if cut < 0 else n[:cut]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--skip", type=int, default=20, help="scans left out at the start (allocation, first-touch)")
    a = ap.parse_args()
    rows = []
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"], short(r["Kernel_Name"])))
    rows.sort()
    # stream roles: by the marker kernels that only ever run on one of them
    marker = {"extract_prepare_kernel": "odometry", "extract_prepare_count_kernel": "odometry", "slam_result_kernel": "mapping (surf side)", "voxel_cloud_lds_kernel<4, 512, false>": "voxel filters"}
    role = {}
    for want in ("odometry", "mapping (surf side)", "voxel filters"):     # markers first: a stream has one role
        for s, e, st, n in rows:
            if marker.get(n) == want and st not in role:
                role[st] = want
                break
    for s, e, st, n in rows:
        if st not in role and n.startswith("grid_"):
            role[st] = "mapping (corner side)"
    # scan boundaries on the odometry stream: the k-th extract_prepare starts scan k; a kernel on another stream belongs to
    # the scan whose chain it is part of — every stream runs the scans in order, so count that stream's own per-scan marker
    per_stream = collections.defaultdict(list)
    for r in rows:
        per_stream[r[2]].append(r)
    first_of_scan = {"odometry": "extract_prepare_kernel", "mapping (surf side)": "slam_map_pose_kernel", "voxel filters": "voxel_cloud_lds_kernel<4, 512, false>",
                     "mapping (corner side)": "grid_mark_kernel"}
    out = []
    scan_span = {}
    for st, lst in per_stream.items():
        name = role.get(st, "stream " + st)
        start_marker = first_of_scan.get(name)
        if start_marker is None:
            continue
        if name == "odometry" and any(r[3] == "extract_prepare_count_kernel" for r in lst):
            start_marker = "extract_prepare_count_kernel"           # round 5: the ring split of a one-scan call is three launches
        scans, cur, seen_pose = [], None, 0
        for r in lst:
            is_start = r[3] == start_marker
            if name == "mapping (surf side)" and is_start:          # two slam_map_pose_kernel per scan: the first one opens it
                seen_pose += 1
                is_start = seen_pose % 2 == 1
            if name == "voxel filters" and is_start:                # two small-form launches per scan (corner list, surf list)
                seen_pose += 1
                is_start = seen_pose % 2 == 1
            if is_start:
                cur = []
                scans.append(cur)
            if cur is not None:
                cur.append(r)
        scans = scans[a.skip:-1] if len(scans) > a.skip + 2 else scans
        if not scans:
            continue
        busy = [sum(e - s for s, e, _, _ in sc) / 1e3 for sc in scans]
        last = {"odometry": "slam_odom_pose_kernel", "mapping (surf side)": "slam_result_kernel"}.get(name)   # what follows opens the next scan
        def end_of(sc):
            ends = [e for _, e, _, n in sc if n == last] if last else []
            return ends[-1] if ends else max(e for _, e, _, _ in sc)
        span = [(end_of(sc) - sc[0][0]) / 1e3 for sc in scans]
        period = [(scans[i + 1][0][0] - scans[i][0][0]) / 1e3 for i in range(len(scans) - 1)]
        kcount = statistics.mean(len(sc) for sc in scans)
        out.append((name, kcount, statistics.median(busy), statistics.median(span), statistics.median(period) if period else float("nan")))
        per_kernel = collections.defaultdict(lambda: [0, 0.0])
        for sc in scans:
            for s, e, _, n in sc:
                per_kernel[n][0] += 1
                per_kernel[n][1] += (e - s) / 1e3
        scan_span[name] = (len(scans), per_kernel)
    print("| stream | kernels per scan | busy, us (median) | span, us (median) | idle inside the span | period, us (median) |")
    print("|---|---|---|---|---|---|")
    for name, kc, b, sp, per in sorted(out):
        print(f"| {name} | {kc:.1f} | {b:.0f} | {sp:.0f} | {100 * (1 - b / sp):.0f} % | {per:.0f} |")
    for name in sorted(scan_span):
        n_sc, pk = scan_span[name]
        print(f"\n**{name}** — per scan:\n")
        print("| kernel | calls | us |")
        print("|---|---|---|")
        for k, (c, t) in sorted(pk.items(), key=lambda kv: -kv[1][1]):
            print(f"| `{k}` | {c / n_sc:.1f} | {t / n_sc:.1f} |")


if __name__ == "__main__":
    sys.exit(main())
