"""Histogram of LM iterations / accepted steps per outer iteration on the bench workload (diagnostic)."""
import json, sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from msf_loam_amd import capi
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
h = capi.Handle(0)
inp = bench.build_inputs(B, 200000, 0, bench.product_extractor(h))
h.set_map(inp["map_corner"], inp["map_surf"])
poses, status, info = h.match_scan2map_batch(inp["corner"], inp["corner_off"], inp["surf"], inp["surf_off"], inp["guesses"].copy(), want_info=True)
out = {}
for it in range(2):
    iters = np.array([i.lm_iterations[it] for i in info]); succ = np.array([i.lm_successful[it] for i in info])
    out[f"outer{it}"] = {"iterations": np.bincount(iters, minlength=8).tolist(), "successful": np.bincount(succ, minlength=8).tolist()}
print(json.dumps(out))
ne = np.array([[i.n_edge[it] for i in info] for it in range(2)]); npl = np.array([[i.n_plane[it] for i in info] for it in range(2)])
nc = np.diff(inp["corner_off"]); ns = np.diff(inp["surf_off"])
print(json.dumps({"edges_accepted_fraction": (ne.sum(1) / nc.sum()).tolist(), "planes_accepted_fraction": (npl.sum(1) / ns.sum()).tolist(),
                  "edges_per_scan": float(nc.mean()), "planes_per_scan": float(ns.mean())}))
