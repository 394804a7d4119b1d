#!/bin/bash
# same-box A/B of the SLAM replay: bash tools/slam_ab.sh "<lib> <lib> ..." "<world>:<beams> ..." [scans]
#   a lib is a path relative to the repo root, or "-" for msf_loam_amd/libmsfl_hip.so
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
LIBS=${1:--}; CASES=${2:-room:16}; N=${3:-60}
for c in $CASES; do
  w=${c%%:*}; b=${c##*:}
  for lib in $LIBS; do
    for m in slam slam-pipelined; do
      if [ "$lib" = "-" ]; then unset MSFL_LIB; else export MSFL_LIB=$R/$lib; fi
      out=$(timeout 300 python $R/examples/replay_synthetic.py --scans $N --beams $b --mode $m --world $w 2>&1 | tail -1)
      echo "$w $b $lib $m $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4f ms/scan  ate %.6g  final %s" % (d["ms_per_scan_end_to_end"], d["ate_rmse_m"], d["final_error_m_rad"]))' 2>/dev/null || echo "$out" | tail -c 300)"
    done
  done
done
