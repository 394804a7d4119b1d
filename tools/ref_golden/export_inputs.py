#!/usr/bin/env python3
"""tests/golden/msfl_golden_v1.npz -> the raw input file of tools/ref_golden/ref_golden (layout: README.md next to this file).
    python tools/ref_golden/export_inputs.py /tmp/msfl_golden_in.bin
Also usable as a module: `write_inputs(path)`, `read_outputs(path)` (the latter is what tests/test_golden.py uses)."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLDEN = os.path.join(ROOT, "tests", "golden", "msfl_golden_v1.npz")
MAGIC_IN, MAGIC_OUT, VERSION = 0x4d53464c, 0x4d534652, 1


def _cloud(f, pts, ring=None):
    pts = np.ascontiguousarray(pts, "<f4").reshape(-1, 4)
    f.write(struct.pack("<i", len(pts)))
    f.write(pts.tobytes())
    if ring is not None:
        f.write(np.ascontiguousarray(ring, "<u2").tobytes())


def write_inputs(path, golden=GOLDEN):
    G = np.load(golden)
    with open(path, "wb") as f:
        f.write(struct.pack("<II", MAGIC_IN, VERSION))
        _cloud(f, G["map_corner"]); _cloud(f, G["map_surf"])
        for k in range(2):
            _cloud(f, G[f"s{k}_corner_ds"]); _cloud(f, G[f"s{k}_surf_ds"])
            f.write(np.asarray(G[f"s{k}_guess"], "<f8").tobytes())
        # scan-to-scan: scan 0's less-sharp / less-flat features are "last", the odo scan's sharp / flat ones "curr".  The current
        # scan's lists are not in the fixture (they are the oracle's extraction of odo_pts): re-derive them with the committed
        # oracle so that the file is self-contained for the reference side.
        sys.path.insert(0, ROOT)
        from oracle import oracle as orc
        orc.build()
        fb = orc.extract_features(G["odo_pts"], G["odo_ring"])
        full0, ring0 = G["s0_full"], G["s0_full_ring"]
        _cloud(f, full0[G["s0_less_sharp"]], ring0[G["s0_less_sharp"]])
        _cloud(f, full0[G["s0_less_flat"]], ring0[G["s0_less_flat"]])
        _cloud(f, fb["full"][fb["sharp"]], fb["ring"][fb["sharp"]])
        _cloud(f, fb["full"][fb["flat"]], fb["ring"][fb["flat"]])
        f.write(np.array([0, 0, 0, 0, 0, 0, 1.0], "<f8").tobytes())
    return path


def read_outputs(path):
    raw = open(path, "rb").read()
    magic, version = struct.unpack_from("<II", raw, 0)
    if magic != MAGIC_OUT or version != VERSION or len(raw) != 8 + 3 * 56 + 4:
        raise ValueError("%s is not a ref_golden output (magic %#x, version %d, %d bytes)" % (path, magic, version, len(raw)))
    poses = np.frombuffer(raw, "<f8", 21, 8).reshape(3, 7)
    (ok,) = struct.unpack_from("<i", raw, 8 + 168)
    return dict(map_poses=poses[:2].copy(), odo_pose=poses[2].copy(), odo_ok=int(ok))


def write_outputs(path, map_poses, odo_pose, odo_ok):
    """The same layout from Python (used by the self-test of the hook: the ORACLE's vectors written as if they were the reference's)."""
    with open(path, "wb") as f:
        f.write(struct.pack("<II", MAGIC_OUT, VERSION))
        f.write(np.asarray(map_poses, "<f8").reshape(2, 7).tobytes())
        f.write(np.asarray(odo_pose, "<f8").reshape(7).tobytes())
        f.write(struct.pack("<i", int(odo_ok)))


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else "msfl_golden_in.bin"
    print("wrote", write_inputs(out), os.path.getsize(out), "bytes")
