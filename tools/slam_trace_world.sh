#!/bin/bash
# kernel trace of the SLAM replay in another world / with another sensor -> per-stream timeline:
#   bash tools/slam_trace_world.sh <room|outdoor|corridor> [scans] [beams]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=${1:-corridor}; N=${2:-80}; B=${3:-16}
O=$R/gpurun_out/tl_${W}_$B; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- python $R/examples/replay_synthetic.py --scans $N --mode slam-pipelined --world $W --beams $B > $O/log.txt 2>&1
f=$(find $O/t -name "t_kernel_trace.csv" | head -1)
python $R/tools/slam_timeline.py $f > $O/slam-pipelined.md
cp $(find $O/t -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/t
cat $O/slam-pipelined.md
