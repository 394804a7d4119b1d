#!/bin/bash
# kernel trace of the SLAM replay -> per-stream timeline (tools/slam_timeline.py); run on the GPU box:
#   gpurun -- 'bash tools/slam_trace.sh [scans]'   -> gpurun_out/tl/{slam,slam-pipelined}.{md,log}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-140}
mkdir -p $R/gpurun_out/tl
cd /tmp && export TMPDIR=/tmp
for m in slam-pipelined slam; do
  python $R/examples/replay_synthetic.py --scans 300 --mode $m 2>&1 | tail -1 | cut -c1-220 > $R/gpurun_out/tl/$m.untraced.log
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl/$m -o t -- python $R/examples/replay_synthetic.py --scans $N --mode $m > $R/gpurun_out/tl/$m.log 2>&1
  f=$(find $R/gpurun_out/tl/$m -name "t_kernel_trace.csv" | head -1)
  python $R/tools/slam_timeline.py $f > $R/gpurun_out/tl/$m.md
  rm -rf $R/gpurun_out/tl/$m
done
cat $R/gpurun_out/tl/*.untraced.log
head -8 $R/gpurun_out/tl/slam-pipelined.md
