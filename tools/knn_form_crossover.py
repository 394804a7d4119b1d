#!/usr/bin/env python3
"""Where the row-parallel (latency) form of the 5-NN kernel stops paying: time of the association launch (event-timed,
msfl_set_timing) for batches of 1 .. 64 scans against the bench map, one-lane-per-query form vs sixteen-lanes-per-query
form (MSFL_KNN_FORM=lane / rows at handle creation).  The automatic rule (kKnnRowsMaxRecords) should sit below the
crossover.  Prints a markdown table; results are also checked to be identical.

    gpurun -- 'python tools/knn_form_crossover.py > gpurun_out/knn_forms.md'
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import bench
    from msf_loam_amd import capi
    torch.zeros(1, device="cuda:0")
    sizes = [1, 2, 4, 6, 8, 12, 16, 24, 32, 64]
    inp = bench.build_inputs(max(sizes), 200000, 0)
    dev = torch.device("cuda", 0)
    d_c, d_s = torch.from_numpy(inp["corner"]).to(dev), torch.from_numpy(inp["surf"]).to(dev)
    hs = {}
    for form in ("lane", "rows"):
        os.environ["MSFL_KNN_FORM"] = form
        hs[form] = capi.Handle(0)
        hs[form].set_map(inp["map_corner"], inp["map_surf"])
    del os.environ["MSFL_KNN_FORM"]
    print("| scans | queries | one lane per query, us per launch | sixteen lanes per query, us per launch | registration call, us (lane / rows) |")
    print("|---|---|---|---|---|")
    for B in sizes:
        co, so = inp["corner_off"][:B + 1], inp["surf_off"][:B + 1]
        row, poses = {}, {}
        for form, h in hs.items():
            for timing in (2, 0):
                h.set_timing(timing)
                ts = []
                for rep in range(12):
                    d_p = torch.from_numpy(inp["guesses"][:B].copy()).to(dev)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    import time
                    t0 = time.perf_counter()
                    h.match_scan2map_batch_device(B, d_c.data_ptr(), co, d_s.data_ptr(), so, d_p.data_ptr())
                    h.synchronize()
                    ts.append((time.perf_counter() - t0) * 1e6)
                if timing == 2:
                    t = h.get_timing()
                    row[form] = 1e3 * t.ms_assoc / max(t.launches_assoc, 1)
                else:
                    row[form + "_call"] = float(np.median(ts[2:]))
                poses[form] = d_p.cpu().numpy()
        assert np.array_equal(poses["lane"], poses["rows"]), B
        print(f"| {B} | {int(co[B] + so[B])} | {row['lane']:.1f} | {row['rows']:.1f} | {row['lane_call']:.0f} / {row['rows_call']:.0f} |")
    for h in hs.values():
        h.close()


if __name__ == "__main__":
    main()
