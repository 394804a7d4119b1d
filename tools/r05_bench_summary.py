"""Prints the round-5 additions of a bench.py JSON line (stdin or file) compactly."""
import json
import sys

d = json.load(open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin)
print("value %.0f  ms/step %.4f  steady %s / %s  prep %.1fs" % (d["value"], d["ms_per_step"], d.get("value_steady"), d.get("ms_per_step_steady"), d["prep_s"]))
print("kernels", d["kernels_ms"])
print("roofline frac", d["roofline"]["frac"], "h2d", (d.get("value_incl_h2d") or {}).get("value"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
st = d.get("stages") or {}
print("stages wall_s", st.get("wall_s"), list(st.keys()))
for k in ("outdoor", "corridor"):
    w = (st.get("worlds") or {}).get(k)
    if w:
        print(k, "reg/s %.0f ms %.3f" % (w["registrations_per_s"], w["ms_per_step"]), w["kernels_ms"], w["knn"], "failed", w["n_failed"], w["oracle_spot_check"],
              "prep %.1f" % w["prep_s"], "vox %.3f ext %.3f" % (w["ms_voxel"], w["ms_extract"]), "F/scan %.0f" % w["features_per_scan"])
for k in ("config3_share", "config4_share"):
    if k in st:
        print(k, {x: st[k][x] for x in ("ms_extract", "ms_voxel", "ms_register_incl_index", "ms_end_to_end", "oracle_spot_check")})
if "pairs" in st:
    print("pairs/s", st["pairs"]["pairs_per_s"], st["pairs"]["oracle_spot_check"])
for k in ("extract", "voxel", "pipeline", "scan2scan", "slam_step"):
    if k in st:
        print(k, {x: v for x, v in st[k].items() if x in ("ms", "scans_per_s", "pairs_per_s", "ms_per_scan_synchronous", "ms_per_scan_pipelined", "frac")})
