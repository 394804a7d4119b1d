"""Diagnostic: does the pipelined SLAM period depend on how many HIP streams the process created before the pipeline's four?
(HIP maps streams onto a small pool of hardware queues round-robin; two streams on one queue serialise.)  GPU box only."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'examples'))
from msf_loam_amd import synth, capi
import replay_synthetic as rp
n = 100
sw = synth.World(ground_half=45.0)
tr = rp.trajectory(120)[:n]
scans = [synth.make_scan(sw, tr[k], synth.SEED + 5000 + k) for k in range(n)]
import gc; gc.collect(); gc.disable()
res = {}
keep = []
for extra in range(0, 9):
    if extra:
        hh = capi.Handle(0); hh.set_map(*synth.make_map(synth.World(ground_half=20.0)))   # a live handle whose stream has been used
        keep.append(hh)
    res[extra] = [round(rp.run_slam(sw, tr, pipelined=True, scans=scans)[2], 3), round(rp.run_slam(sw, tr, pipelined=False, scans=scans)[2], 3)]
print(json.dumps(res))
