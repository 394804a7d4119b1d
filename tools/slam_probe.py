"""Diagnostic: wall clock per scan of the SLAM step under different host-process conditions (torch imported, the oracle's OpenMP
pool warmed up) -- the step is host-enqueue bound, so anything that slows the calling thread shows.  GPU box only."""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'examples'))
import numpy as np
from msf_loam_amd import synth
import replay_synthetic as rp
mode = sys.argv[1:] 
if 'torch' in mode:
    import torch
    torch.zeros(1, device='cuda')
if 'omp' in mode:       # a 256-thread OpenMP region in this process first (what bench.py's cpu_baseline leg leaves behind)
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, 'oracle', 'libmsfl_oracle.so'))
    from msf_loam_amd import capi
    w = synth.World(ground_half=synth.ground_half_for_target(30000))
    mc, ms = synth.make_map(w)
    sys.path.insert(0, ROOT)
    import importlib
    orc = importlib.import_module('oracle.oracle')
    p = synth.random_poses(8, 5)
    fe = [synth.direct_features(*[synth.make_scan(w, p[i], 7 + i, with_kind=True)[j] for j in (0, 2)]) for i in range(8)]
    co = np.cumsum([0] + [len(c) for c, _ in fe]).astype(np.int32); so = np.cumsum([0] + [len(s) for _, s in fe]).astype(np.int32)
    orc.match_scan2map_batch(mc, ms, np.concatenate([c for c, _ in fe]), co, np.concatenate([s for _, s in fe]), so, p, threads=os.cpu_count())
if 'nullstream' in mode or 'pin' in mode or 'events' in mode:
    import torch
    import bench
    from msf_loam_amd import capi
    dev = torch.device('cuda', 0)
    hb = capi.Handle(0)
    inp = bench.build_inputs(64, 50000, 0, bench.product_extractor(hb))
    if 'nullstream' in mode:
        hb.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    d = {k: torch.from_numpy(inp[k]).to(dev) for k in ('map_corner', 'map_surf', 'corner', 'surf', 'guesses')}
    dp = torch.zeros((64, 7), dtype=torch.float64, device=dev); ds = torch.zeros(64, dtype=torch.int32, device=dev)
    if 'events' in mode:
        hb.set_timing(2)
    for _ in range(5):
        dp.copy_(d['guesses'])
        hb.set_map(d['map_corner'], d['map_surf'], len(inp['map_corner']), len(inp['map_surf']), capi.MEM_DEVICE)
        hb.match_scan2map_batch_device(64, d['corner'], inp['corner_off'], d['surf'], inp['surf_off'], dp, ds)
    torch.cuda.synchronize()
    if 'events' in mode:
        hb.get_timing(reset=True); hb.set_timing(0)
    if 'pin' in mode:
        bench.host_buffer_rate(hb, inp, 64)
n = 100
sw = synth.World(ground_half=45.0)
tr = rp.trajectory(120)[:n]
scans = [synth.make_scan(sw, tr[k], synth.SEED + 5000 + k) for k in range(n)]
import gc; gc.collect(); gc.disable()
res = {}
for name, pl in (("pipelined_first", True), ("sync", False), ("pipelined_again", True)):
    res[name] = round(rp.run_slam(sw, tr, pipelined=pl, scans=scans)[2], 4)
print(mode, json.dumps(res))
