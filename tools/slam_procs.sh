#!/bin/bash
# K concurrent PROCESSES, one SLAM replay each, on one GPU: bash tools/slam_procs.sh <K> [scans] [world] [beams]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
K=${1:-4}; N=${2:-300}; W=${3:-room}; B=${4:-16}
T0=$(date +%s.%N)
for i in $(seq 1 $K); do
  python $R/examples/replay_synthetic.py --scans $N --mode slam-pipelined --world $W --beams $B > /tmp/slam_proc_$i.json 2>/dev/null &
done
wait
T1=$(date +%s.%N)
python - <<PY
import json
ms=[json.loads(open('/tmp/slam_proc_%d.json'%i).read().strip().splitlines()[-1])['ms_per_scan_end_to_end'] for i in range(1,$K+1)]
print(json.dumps({"processes": $K, "scans_each": $N, "world": "$W", "beams": $B, "ms_per_scan_in_each_process": [round(m,4) for m in ms],
  "scans_per_s_all_processes_steady": sum(1e3/m for m in ms), "wall_s_incl_start_up_and_scan_generation": $T1-$T0}))
PY
