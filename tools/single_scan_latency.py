"""Latency of one msfl_match_scan2map call (host buffers, map resident): python tools/single_scan_latency.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from msf_loam_amd import capi
h = capi.Handle(0)
inp = bench.build_inputs(8, 200000, 0, extractor=bench.product_extractor(h))
h.set_map(inp["map_corner"], inp["map_surf"])
co, so = inp["corner_off"], inp["surf_off"]
out = {}
for B in (1, 8):
    c, s, g = inp["corner"][:co[B]], inp["surf"][:so[B]], inp["guesses"][:B]
    for _ in range(5):
        h.match_scan2map_batch(c, co[:B + 1], s, so[:B + 1], g)
    t0 = time.perf_counter()
    K = 50
    for _ in range(K):
        h.match_scan2map_batch(c, co[:B + 1], s, so[:B + 1], g)
    out["B=%d_ms_per_call" % B] = 1e3 * (time.perf_counter() - t0) / K
print(json.dumps(out))
