"""Profile build only (make gridprof; MSFL_LIB=<that library>): per-cell phase times of the map store's rebuild kernels for the last insert
of a short SLAM replay.  python tools/grid_prof.py [world] [beams] [scans]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "examples"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import replay_synthetic as rp
from msf_loam_amd import capi

world, beams, n = (sys.argv[1:] + ["room", "64", "8"])[:3]
from msf_loam_amd import synth
n = int(n)
if world == "room":
    w = synth.World(ground_half=45.0); truth = rp.trajectory(n)
else:
    w = synth.World(kind=world); truth = rp.world_drive(w, world, n)
kw = dict(n_beams=64, n_az=1900, elev=(-24.8, 2.0)) if int(beams) == 64 else {}
scans = [synth.make_scan(w, truth[k], synth.SEED + 5000 + k, **kw) for k in range(n)]
rp.run_slam(w, truth, pipelined=False, scans=scans)
lib = capi.load()
buf = np.zeros(1024 * 8, np.uint64)
rc = lib.msfl_debug_grid_prof(buf.ctypes.data_as(C.c_void_p), C.c_int(len(buf)))
q = buf.reshape(1024, 8)
q = q[q[:, 0] != 0]
E = (q[:, 0] & 0xffffffff).astype(int); thr = (q[:, 0] >> 32).astype(int)
t0 = q[:, 5].min()
order = np.argsort(-E)
print("rc", rc, "cells", len(q), "kernel window (us):", (q[:, 6].max() - t0) / 100.0)
print("    E thr  fill  sort heads  long | start   end (us)  block")
for i in list(order[:12]) + list(order[len(order) // 2: len(order) // 2 + 4]):
    print("%5d %4d %5.1f %5.1f %5.1f %5.1f | %6.1f %6.1f  %d" % (E[i], thr[i], q[i, 1] / 100, q[i, 2] / 100, q[i, 3] / 100, q[i, 4] / 100, (q[i, 5] - t0) / 100, (q[i, 6] - t0) / 100, q[i, 7] >> 32))

vb = np.zeros(64 * 8, np.uint64)
if hasattr(lib, "msfl_debug_vox_prof") and lib.msfl_debug_vox_prof(vb.ctypes.data_as(C.c_void_p), C.c_int(len(vb))) == 0:
    v = vb.reshape(64, 8)
    print("voxel filter, last launches (us): form  points -> voxels  runs multi big | set-up+phase1  phase2  sort  heads  threads  big-voxel wavefronts")
    for i in range(24):
        if v[i, 6] == 0:
            continue
        form = ("<4,512>", "<16,768>", "<16,4096,global>")[i // 8]
        print("  %-17s %6d -> %5d  %5d %4d %4d | %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f" % (form, v[i, 6] >> 32, v[i, 6] & 0xffffffff, v[i, 7] >> 32, (v[i, 7] >> 16) & 0xffff, v[i, 7] & 0xffff,
                                                                            v[i, 0] / 100, v[i, 1] / 100, v[i, 2] / 100, v[i, 3] / 100, v[i, 4] / 100, v[i, 5] / 100))
