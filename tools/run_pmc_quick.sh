#!/bin/bash
# One or two counter passes only (traffic + instruction mix) over a bench.py command line: a quick look between kernel changes.
#   gpurun -- 'bash tools/run_pmc_quick.sh TAG "--genome realistic --cpu-sample 0 --ref-sample 0 --steps 2"'
set -e
TAG=${1:-q}
ARGS=${2:---cpu-sample 0 --ref-sample 0 --steps 2}
export BASAL_BENCH_NO_H2H=1
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
GROUPS_=(
  "TCC_HIT_sum TCC_MISS_sum"
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
)
i=0
for g in "${GROUPS_[@]}"; do
  timeout -k 10 400 rocprofv3 --pmc $g --output-format csv -d "$OUT/pmc$i" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc$i.json" 2> "$OUT/pmc$i.err" || echo "[pmc] group $i failed (see pmc$i.err)"
  i=$((i+1))
done
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT" > "gpurun_out/pmc_$TAG.csv"
cat "gpurun_out/pmc_$TAG.csv"
find "$OUT" -name "*counter_collection.csv" -delete
