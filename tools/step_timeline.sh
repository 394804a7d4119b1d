#!/bin/bash
# per-kernel timeline of the bench step (start offsets, durations, gaps): bash tools/step_timeline.sh <tag>     (on the GPU box)
TAG=${1:-tl_step}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o t -- python $ROOT/bench.py --steps 12 --warmup 3 --no-h2d --no-stages --no-worlds --cpu-sample 0 --steady-steps 0 --no-kernel-timing > $OUT/log.txt 2>&1
K=$(find $OUT -name "*kernel_trace.csv" | head -1); M=$(find $OUT -name "*memory_copy_trace.csv" | head -1)
python - "$K" "$M" > $OUT/step_timeline.md <<'PY'
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
try:
    for r in csv.DictReader(open(sys.argv[2])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") ))
except Exception as e:
    pass
ev.sort()
# steps: from one grid_bbox_kernel of the corner map to the next pair; find lm_solve occurrences and take the last 5 steps
idx = [i for i, e in enumerate(ev) if e[2].startswith("void msfl::lm_solve_kernel<128>") or "lm_solve_kernel<128>" in e[2]]
# a step has two solves: cut after every second
cuts = idx[1::2]
print("# bench step timeline (rocprofv3 --kernel-trace --memory-copy-trace), last full steps\n")
for s in range(len(cuts) - 4, len(cuts) - 1):
    a, b = cuts[s] + 1, cuts[s + 1] + 1
    t0 = ev[a][0]
    prev_end = ev[cuts[s]][1]
    print("| start us | dur us | gap before us | what |\n|---|---|---|---|")
    tot_gap = 0.0
    for e in ev[a:b]:
        gap = (e[0] - prev_end) / 1e3
        tot_gap += max(gap, 0)
        print("| %8.1f | %7.1f | %6.1f | %s |" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, gap, e[2]))
        prev_end = max(prev_end, e[1])
    print("\nstep span %.1f us, sum of gaps %.1f us\n" % ((ev[b - 1][1] - ev[cuts[s]][1]) / 1e3, tot_gap))
PY
cat $OUT/step_timeline.md | tail -40
find $OUT -type f ! -name step_timeline.md ! -name log.txt -delete
