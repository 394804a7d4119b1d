"""K independent SLAM sessions (msfl_slam objects) on ONE GPU, one host thread each: what `replicas only` buys on a
single device.  The per-scan step is a chain of ~75 dependent launches that keeps a few of the 256 compute units busy; K sessions are K
such chains on their own streams.  Every session replays the same drive, so every pose track must equal the single session's bit for bit.
    python tools/slam_sessions.py [scans] [world] [beams] [K ...]        -> one JSON line"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
import replay_synthetic as rp  # noqa: E402
from msf_loam_amd import synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    world = sys.argv[2] if len(sys.argv) > 2 else "room"
    beams = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    ks = [int(a) for a in sys.argv[4:]] or [1, 2, 4, 8, 16]
    if world == "room":
        w = synth.World(ground_half=45.0); truth = rp.trajectory(n)
    else:
        w = synth.World(kind=world); truth = rp.world_drive(w, world, n)
    kw = dict(n_beams=64, n_az=1900, elev=(-24.8, 2.0)) if beams == 64 else {}
    scans = [synth.make_scan(w, truth[k], synth.SEED + 5000 + k, **kw) for k in range(n)]
    ref, _, _ = rp.run_slam(w, truth, pipelined=True, scans=scans)            # also warms the library up
    out = {"scans_per_session": n, "world": world, "beams": beams, "sessions": {}}
    for K in ks:
        est = [None] * K
        ms = [0.0] * K
        gate = threading.Barrier(K + 1)

        def work(i):
            gate.wait()
            est[i], _, ms[i] = rp.run_slam(w, truth, pipelined=True, scans=scans)

        th = [threading.Thread(target=work, args=(i,)) for i in range(K)]
        for t in th:
            t.start()
        gate.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        wall = time.perf_counter() - t0
        same = all(np.array_equal(e, ref) for e in est)
        out["sessions"][str(K)] = {"scans_per_s_all_sessions": K * n / wall, "ms_per_scan_per_session": float(np.mean(ms)),
                                   "wall_s": wall, "poses_equal_the_single_session_bitwise": bool(same)}
    out["note"] = ("wall clock from the common start to the last session's end, session set-up (allocations, first scans) included; "
                   "Python threads (ctypes releases the GIL inside the calls)")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
