"""PCIe-inclusive rate of the batch registration: features, guesses and results in (pageable / pinned) host memory,
map resident.  python tools/host_buffer_rate.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from msf_loam_amd import capi

torch.zeros(1, device="cuda:0")
h = capi.Handle(0)
inp = bench.build_inputs(1024, 200000, 0, extractor=bench.product_extractor(h))
h.set_map(inp["map_corner"], inp["map_surf"])
co, so = inp["corner_off"], inp["surf_off"]
import gc
gc.collect(); gc.disable()
res = {}
for label, pin in (("pageable", False), ("pinned", True)):
    c, s, g = inp["corner"], inp["surf"], inp["guesses"]
    if pin:
        c, s, g = (torch.from_numpy(a).pin_memory().numpy() for a in (c, s, g))
    for _ in range(3):
        h.match_scan2map_batch(c, co, s, so, g)
    t0 = time.perf_counter()
    K = 10
    for _ in range(K):
        poses, status, _ = h.match_scan2map_batch(c, co, s, so, g)
    dt = (time.perf_counter() - t0) / K
    res[label] = {"ms_per_batch": 1e3 * dt, "registrations_per_s": 1024 / dt, "host_bytes_in": int(c.nbytes + s.nbytes + g.nbytes)}
print(json.dumps(res))
