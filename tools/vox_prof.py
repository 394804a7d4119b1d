"""Profile build only (make gridprof; MSFL_LIB=<that library>): phase clocks of the voxel filter's one-workgroup forms on a batch of less-flat lists.
    python tools/vox_prof.py [beams] [scans]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from msf_loam_amd import capi, synth

beams, n = (sys.argv[1:] + ["64", "8"])[:2]
beams, n = int(beams), int(n)
w = synth.World(ground_half=45.0)
poses = synth.random_poses(n, synth.SEED + 9)
kw = dict(n_beams=64, n_az=1900, elev=(-24.8, 2.0)) if beams == 64 else {}
h = capi.Handle(0)
clouds = []
for k in range(n):
    pts, ring = synth.make_scan(w, poses[k], synth.SEED + 700 + k, **kw)[:2]
    f = h.extract_features(pts, ring)
    clouds.append(f["full"][f["less_flat"]])
off = np.cumsum([0] + [len(c) for c in clouds]).astype(np.int32)
out, out_off = h.voxel_downsample_batch(np.concatenate(clouds), off, 0.4)
lib = capi.load()
vb = np.zeros(64 * 8, np.uint64)
assert lib.msfl_debug_vox_prof(vb.ctypes.data_as(C.c_void_p), C.c_int(len(vb))) == 0
v = vb.reshape(64, 8)
print("form  points -> voxels  runs multi big | set-up+phase1  phase2  sort  heads  threads  big-voxel wavefronts (us)")
for i in range(24):
    if v[i, 6] == 0:
        continue
    form = ("<4,512>", "<16,768>", "<16,4096,global/big>")[i // 8]
    print("  %-21s %6d -> %5d  %5d %4d %4d | %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f" % (form, v[i, 6] >> 32, v[i, 6] & 0xffffffff, v[i, 7] >> 32, (v[i, 7] >> 16) & 0xffff, v[i, 7] & 0xffff,
                                                                          v[i, 0] / 100, v[i, 1] / 100, v[i, 2] / 100, v[i, 3] / 100, v[i, 4] / 100, v[i, 5] / 100))
h.close()
