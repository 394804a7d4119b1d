"""The bench step (pose copy + msfl_set_map + msfl_match_scan2map_batch on device-resident inputs) eager vs replayed from a captured HIP
graph (round 6; VERDICT r04 weak #10 asked for a captured graph instead of an argument).  python tools/graph_step.py [scans] [steps]
Prints one JSON line: ms per step eager / graph, and whether the poses of a replay equal the eager ones bit for bit."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from msf_loam_amd import capi  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
h = capi.Handle(0)
inp = bench.build_inputs(B, 200000, 0, bench.product_extractor(h), with_map=True)
s = torch.cuda.Stream(dev)
h.set_stream(s.cuda_stream)
with torch.cuda.stream(s):
    d_map_c = torch.from_numpy(inp["map_corner"]).to(dev); d_map_s = torch.from_numpy(inp["map_surf"]).to(dev)
    d_corner = torch.from_numpy(inp["corner"]).to(dev); d_surf = torch.from_numpy(inp["surf"]).to(dev)
    d_guess = torch.from_numpy(inp["guesses"]).to(dev)
    d_poses = torch.zeros((B, 7), dtype=torch.float64, device=dev); d_status = torch.zeros(B, dtype=torch.int32, device=dev)
n_mc, n_ms = len(inp["map_corner"]), len(inp["map_surf"])
co, so = inp["corner_off"], inp["surf_off"]


def step():
    d_poses.copy_(d_guess)
    h.set_map(d_map_c, d_map_s, n_mc, n_ms, capi.MEM_DEVICE)
    h.match_scan2map_batch_device(B, d_corner, co, d_surf, so, d_poses, d_status)


def timed(fn, n):
    for _ in range(10):
        fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(dev)
    return 1e3 * (time.perf_counter() - t0) / n


import gc
gc.collect(); gc.disable()
with torch.cuda.stream(s):
    for _ in range(5):
        step()                                  # every buffer allocated, table spans settled, offsets cached: nothing but launches left
    torch.cuda.synchronize(dev)
    eager_a = timed(step, steps)
    ref = d_poses.cpu().numpy().copy()
g = torch.cuda.CUDAGraph()
err = None
try:
    with torch.cuda.graph(g, stream=s):
        step()
    torch.cuda.synchronize(dev)
    graph_ms = timed(g.replay, steps)
    same = bool(np.array_equal(d_poses.cpu().numpy(), ref))
    with torch.cuda.stream(s):
        eager_b = timed(step, steps)
    graph_b = timed(g.replay, steps)
except Exception as e:      # noqa: BLE001
    err = repr(e)[:400]; graph_ms = graph_b = eager_b = None; same = None
print(json.dumps({"scans": B, "steps": steps, "ms_per_step_eager": [eager_a, eager_b], "ms_per_step_graph_replay": [graph_ms, graph_b],
                  "replay_poses_equal_eager_bitwise": same, "error": err,
                  "note": "one stream; the graph holds the pose copy, the 5 index-build launches, the status memset and the 6 registration kernels"}))
