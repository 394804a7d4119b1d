"""Device-resident batch pipeline over the C ABI: raw scans -> feature extraction (stage A) -> the two voxel filters
(laser_mapping.cc:264-270) -> scan-to-map registration (stage C), B scans per call, nothing but offsets and counts
visiting the host.  Host-side harness for tools/, bench.py and the per-GPU-share tests of BASELINE configs[3] / [4];
the compute is libmsfl_hip.so's (no CPU path)."""
import ctypes as C

import numpy as np
import torch

from . import capi


class BatchPipeline:
    def __init__(self, handle, pts, ring, off, device=None):
        """pts (n,4) f32, ring (n,) u16, off (B+1,) i32: B scans concatenated in driver order."""
        self.h, self.lib = handle, handle.lib
        self.dev = device or torch.device("cuda", 0)
        self.off = np.ascontiguousarray(off, dtype=np.int32)
        self.B, n = len(self.off) - 1, int(self.off[-1])
        dev = self.dev
        self.d_pts = torch.from_numpy(np.ascontiguousarray(pts, dtype=np.float32)).to(dev)
        self.d_ring = torch.from_numpy(np.ascontiguousarray(ring).astype(np.int16)).to(dev)
        self.d_full = torch.empty((n, 4), dtype=torch.float32, device=dev)
        self.d_fring = torch.empty(n, dtype=torch.int16, device=dev)
        self.d_curv = torch.empty(n, dtype=torch.float32, device=dev)
        self.d_label = torch.empty(n, dtype=torch.uint8, device=dev)
        self.d_idx = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(4)]       # sharp, less_sharp, flat, less_flat
        self.d_cnt = [torch.empty(self.B, dtype=torch.int32, device=dev) for _ in range(5)]  # full + the four lists
        self.d_status = torch.empty(self.B, dtype=torch.int32, device=dev)
        self.d_corner = torch.empty((n, 4), dtype=torch.float32, device=dev)
        self.d_surf = torch.empty((n, 4), dtype=torch.float32, device=dev)
        self.corner_off = np.zeros(self.B + 1, np.int32)
        self.surf_off = np.zeros(self.B + 1, np.int32)
        f = capi.FeaturesBatch()
        f.full_pts, f.full_ring, f.curvature, f.label = self.d_full.data_ptr(), self.d_fring.data_ptr(), self.d_curv.data_ptr(), self.d_label.data_ptr()
        f.sharp_idx, f.less_sharp_idx, f.flat_idx, f.less_flat_idx = (t.data_ptr() for t in self.d_idx)
        f.n_full, f.n_sharp, f.n_less_sharp, f.n_flat, f.n_less_flat = (t.data_ptr() for t in self.d_cnt)
        self.f = f
        self.d_poses = self.d_mstat = None

    def _check(self, s, what):
        if s != 0:
            raise capi.MsflError(s, what, self.lib.msfl_last_error(self.h.h).decode())

    def extract(self):
        vp = C.c_void_p
        self._check(self.lib.msfl_extract_features_batch(self.h.h, C.c_int(self.B), vp(self.d_pts.data_ptr()), vp(self.d_ring.data_ptr()),
                                                         self.off.ctypes.data_as(vp), C.byref(self.f), vp(self.d_status.data_ptr()),
                                                         C.c_int(capi.MEM_DEVICE)), "msfl_extract_features_batch")

    def voxel(self, leaf_corner=0.2, leaf_surf=0.4):
        """corner and surf lists in one call: both filters are enqueued before the one synchronisation."""
        vp = C.c_void_p
        self._check(self.lib.msfl_voxel_downsample_batch_pair(
            self.h.h, C.c_int(self.B), vp(self.d_full.data_ptr()), self.off.ctypes.data_as(vp),
            vp(self.d_idx[1].data_ptr()), vp(self.d_cnt[2].data_ptr()), C.c_float(leaf_corner), vp(self.d_corner.data_ptr()), self.corner_off.ctypes.data_as(vp),
            vp(self.d_idx[3].data_ptr()), vp(self.d_cnt[4].data_ptr()), C.c_float(leaf_surf), vp(self.d_surf.data_ptr()), self.surf_off.ctypes.data_as(vp),
            C.c_int(capi.MEM_DEVICE)), "msfl_voxel_downsample_batch_pair")

    def set_map(self, map_corner, map_surf):
        self.d_map_c = torch.from_numpy(np.ascontiguousarray(map_corner, dtype=np.float32)).to(self.dev)
        self.d_map_s = torch.from_numpy(np.ascontiguousarray(map_surf, dtype=np.float32)).to(self.dev)

    def register(self, d_guess):
        """index the map (per batch, like the kd-tree build of mapping_scan_matcher.cc:66-73) and register all B scans"""
        if self.d_poses is None:
            self.d_poses = torch.empty_like(d_guess)
            self.d_mstat = torch.zeros(self.B, dtype=torch.int32, device=self.dev)
        self.d_poses.copy_(d_guess)
        self.h.set_map(self.d_map_c, self.d_map_s, len(self.d_map_c), len(self.d_map_s), capi.MEM_DEVICE)
        self.h.match_scan2map_batch_device(self.B, self.d_corner, self.corner_off, self.d_surf, self.surf_off, self.d_poses, self.d_mstat)

    def run(self, d_guess):
        self.extract(); self.voxel(); self.register(d_guess)
        return self.d_poses, self.d_mstat
