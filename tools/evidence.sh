#!/bin/bash
TAG=${1:-r05}
# Evidence run on the GPU box (one gpurun call): bash tools/evidence.sh <tag>: full GPU test suite, the bench line, kernel stats, PMC passes,
# SLAM replay variants + timeline.  Outputs under gpurun_out/${TAG}/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${TAG}; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 > $O/gpu_tests.txt
[ -n "${FUZZ:-}" ] && (MSFL_FUZZ_SEEDS=$FUZZ timeout 1500 python -m pytest tests/test_gpu_extract.py tests/test_gpu_scan2map.py tests/test_gpu_scan2scan.py tests/test_grid_store.py tests/test_deskew.py -q -m gpu 2>&1 | tail -4 > $O/fuzz.txt)
timeout 600 python bench.py --steps 200 --warmup 10 > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_shape.json 2>> $O/bench.err
# (before the PMC passes: counter collection leaves the device in its profiling clock state, which cost the pipelined replays 20-25 %)
# round 5b: the step with the 64-beam sensor and in the other worlds: per-stream timelines + 60-scan replays of this build
for c in "room 40 64" "outdoor 40 64" "corridor 80 16" "outdoor 80 16"; do
  set -- $c; timeout 300 bash tools/slam_trace_world.sh $c > /dev/null 2>&1; cp $R/gpurun_out/tl_$1_$3/slam-pipelined.md $O/slam_timeline_$1_$3.md 2>/dev/null
done
timeout 900 bash tools/slam_ab.sh "-" "room:16 room:64 corridor:16 outdoor:16 outdoor:64" > $O/slam_replays.txt 2>&1
timeout 400 bash tools/prof.sh ${TAG}/prof > $O/prof.txt 2>&1
# round 5: the 64-beam share of configs[3] (kernel trace + PMC) and the two other worlds
timeout 400 bash tools/prof_cmd.sh ${TAG}/prof64 python $R/tools/r05_share_ab.py x config3_share > $O/prof64.txt 2>&1
PMC_CMD="python $R/tools/r05_share_ab.py x config3_share" timeout 900 bash tools/pmc.sh ${TAG}/pmc64 > $O/pmc64.txt 2>&1
timeout 400 bash tools/prof_cmd.sh ${TAG}/prof_worlds python $R/tools/bench_worlds.py worlds 256 > $O/prof_worlds.txt 2>&1
PMC_CMD="python $R/tools/bench_worlds.py worlds 256 outdoor" timeout 900 bash tools/pmc.sh ${TAG}/pmc_outdoor > $O/pmc_outdoor.txt 2>&1
timeout 900 bash tools/pmc.sh ${TAG}/pmc > $O/pmc.txt 2>&1
for m in "" "--imu" "--reference-quirks" "--imu --reference-quirks"; do
  for mode in slam slam-pipelined; do timeout 200 python examples/replay_synthetic.py --scans 300 --mode $mode $m 2>/dev/null | tail -1; done
done > $O/replay300.jsonl
MSFL_SLAM_HOST_PROFILE=1 timeout 200 python examples/replay_synthetic.py --scans 300 --mode slam-pipelined > /dev/null 2> $O/slam_host_profile.txt
timeout 600 bash tools/slam_trace.sh 140 > $O/slam_trace.txt 2>&1; cp $R/gpurun_out/tl/*.md $O/ 2>/dev/null
ls -la $O
