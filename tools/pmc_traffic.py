"""profiles/rNN_pmc_summary.txt -> profiles/pmc_traffic.json: HBM bytes per launch of every kernel,
(2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE reports half of wide reads, MI355X_MICROARCH.md
HBM section).  bench.py reads the result for `roofline.traffic`.
    python tools/pmc_traffic.py profiles/r01d_pmc_summary.txt"""
import json
import os
import sys

src = sys.argv[1]
kernels, cur = {}, None
for line in open(src):
    if not line.startswith(" "):
        cur = line.strip()
        kernels[cur] = {}
    else:
        parts = line.split()
        if parts[0] in ("FETCH_SIZE", "WRITE_SIZE"):
            kernels[cur]["fetch_kb" if parts[0] == "FETCH_SIZE" else "write_kb"] = float(parts[2])
        elif parts[0] in ("SQ_INSTS_VALU", "SQ_WAVES", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
                          "SQ_INSTS_LDS", "SQ_INSTS_SALU", "TA_BUSY_avr", "TA_BUSY_max", "GRBM_GUI_ACTIVE", "TCP_TOTAL_CACHE_ACCESSES_sum",
                          "TCP_TCC_READ_REQ_sum"):
            kernels[cur][parts[0].lower()] = float(parts[2])
out = {"source": f"{src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 "
                 "per MI355X_MICROARCH.md HBM note)", "kernels": {}}
for k, v in kernels.items():
    if "fetch_kb" in v and "write_kb" in v:
        out["kernels"][k] = dict(v, hbm_bytes_per_launch=(2 * v["fetch_kb"] + v["write_kb"]) * 1024)
path = os.path.join(os.path.dirname(os.path.abspath(src)), "pmc_traffic.json")
json.dump(out, open(path, "w"), indent=1)
print(path, len(out["kernels"]), "kernels")
