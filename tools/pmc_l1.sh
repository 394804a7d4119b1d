#!/bin/bash
# L1 / texture-addresser counters of the bench workload (one rocprofv3 pass per counter set): bash tools/pmc_l1.sh <tag>
TAG=${1:-pmc_l1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
CMD="python $ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-h2d --no-stages"
i=0
for SET in "TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TA_BUFFER_WAVEFRONTS_sum" \
           "SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_IFETCH SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT -o pass$i -- $CMD > $OUT/pass$i.log 2>&1
done
python $ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -type f ! -name summary.txt ! -name "*.log" ! -name counters.txt -delete
grep -A40 "knn5_scan2map_split" $OUT/summary.txt | head -60
