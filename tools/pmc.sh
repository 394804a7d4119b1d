#!/bin/bash
# (every pass under its own timeout: a counter set the box cannot collect must not take the whole call with it — the
# TA_*_STALLED_* set hung rocprofv3 for 15 minutes in round 4)
# Collect rocprofv3 PMC counters for the bench workload in separate passes (gpurun forbids mixing
# --pmc with tracing domains other than --kernel-trace/--stats).  Run on the GPU box:
#   bash tools/pmc.sh <tag> [bench args]
set -u
TAG=${1:-pmc}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --cpu-sample 0 --no-h2d --no-stages $*"     # --no-h2d: the host-buffer leg launches the kernels over chunks of the batch, which would lower the per-launch means
# PMC_CMD overrides the profiled command (default: the bench workload), e.g. PMC_CMD="python tools/stage_throughput.py 1024"
CMD=${PMC_CMD:-"python $ROOT/bench.py $ARGS"}
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" \
           "WRITE_SIZE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT -o pass$i -- $CMD > $OUT/pass$i.log 2>&1
done
python $ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
du -sh $OUT/* | sort -h | tail -5
find $OUT -type f ! -name summary.txt ! -name "*.log" -delete
cat $OUT/summary.txt
