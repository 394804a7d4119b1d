#!/bin/bash
# kernel-trace summary of a short bench run on the GPU box: bash tools/prof.sh <tag> [extra bench args]
TAG=${1:-prof}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $ROOT/bench.py --steps 20 --warmup 3 --no-h2d --no-stages --cpu-sample 0 "$@" > $OUT/log.txt 2>&1
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print("%-90s calls %6s avg %10.1f us  total %6.2f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
cp "$F" $OUT/kernel_stats.csv; find $OUT -type f ! -name kernel_stats.csv ! -name log.txt -delete
