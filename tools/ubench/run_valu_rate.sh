#!/bin/bash
# On the GPU box: bash tools/ubench/run_valu_rate.sh   -> gpurun_out/valu_rate/{table.txt,valu_rate.json,pmc.txt}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/valu_rate
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -x $ROOT/tools/ubench/valu_rate ] || hipcc --offload-arch=gfx950 -O3 -w $ROOT/tools/ubench/valu_rate.hip -o $ROOT/tools/ubench/valu_rate
$ROOT/tools/ubench/valu_rate $OUT/valu_rate.json > $OUT/table.txt 2>&1
# counters for kernels whose instruction count is known exactly (32 * iters per wavefront + ~40 of prologue)
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc -o p -- \
  $ROOT/tools/ubench/valu_rate $OUT/pmc_run.json pmc > $OUT/pmc.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
rows = defaultdict(dict)
for f in glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        key = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0], int(r.get("Grid_Size", 0) or 0), int(r.get("Workgroup_Size", 0) or 0))
        rows[key][r["Counter_Name"]] = float(r["Counter_Value"])
with open(os.path.join(out, "pmc.txt"), "w") as fo:
    fo.write("dispatch kernel grid wg | SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU  INSTS/wave  ACTIVE/INSTS  SQ_BUSY_CYCLES SQ_WAVE_CYCLES\n")
    for key in sorted(rows):
        c = rows[key]
        w, i, a = c.get("SQ_WAVES", 0), c.get("SQ_INSTS_VALU", 0), c.get("SQ_ACTIVE_INST_VALU", 0)
        fo.write("%4d %-40s %8d %5d | %8.0f %14.0f %14.0f %10.1f %8.3f %14.0f %14.0f\n" % (key[0], key[1][-40:], key[2], key[3], w, i, a, i / w if w else 0,
                 a / i if i else 0, c.get("SQ_BUSY_CYCLES", 0), c.get("SQ_WAVE_CYCLES", 0)))
print(open(os.path.join(out, "pmc.txt")).read()[:6000])
PY
rm -rf $OUT/pmc
cat $OUT/table.txt
