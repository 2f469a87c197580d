#!/bin/bash
# Collect the hardware counters behind bench.py's roofline/traffic figures on the GPU box:
#   gpurun -- 'bash tools/run_pmc.sh r01'     -> gpurun_out/pmc_<tag>/{stats,pmc*}/... and gpurun_out/pmc_<tag>.csv
# One rocprofv3 pass per counter group (PMC passes never combined with tracing, one program after `--`).
# The summary (tools/pmc_summary.py) is what gets copied to profiles/.
export BASAL_BENCH_PLACEMENT_DRAWS=1  # (one placement: the calibration launches of bench.py's placement draws would be averaged into the per-launch figures)
set -e
TAG=${1:-run}
ARGS=${2:---cpu-sample 0 --ref-sample 0 --steps 4}
export BASAL_BENCH_NO_H2H=1
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
GROUPS_=(
  "FETCH_SIZE"
  "WRITE_SIZE"
  "TCC_HIT_sum TCC_MISS_sum"
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY"
  "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"
)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.json" 2> "$OUT/stats.err"
echo "[pmc] kernel-trace pass done"
i=0
for g in "${GROUPS_[@]}"; do
  timeout -k 10 400 rocprofv3 --pmc $g --output-format csv -d "$OUT/pmc$i" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc$i.json" 2> "$OUT/pmc$i.err" || echo "[pmc] group $i failed (see pmc$i.err)"
  echo "[pmc] counter pass $i done"
  i=$((i+1))
done
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT" > "gpurun_out/pmc_$TAG.csv"
cat "gpurun_out/pmc_$TAG.csv"
# the raw per-dispatch tables (every torch kernel of the workload generator x every counter) are tens of MB: keep the summary
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*kernel_trace.csv" -delete
