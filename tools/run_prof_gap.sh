#!/bin/bash
# Diagnostic builds (make prof / make variant V=cnt EXTRA="-DBASAL_PHASE_TIMING -DBASAL_COUNT_ADDHIT") on configs 4 and 5p, hg38-like genome:
# phase clocks, survivor counts and what AddHit's calls end in. Output under gpurun_out/.
set -e
cd "$(dirname "$0")/.."
export BASAL_BENCH_NO_H2H=1
for cfg in ${CFGS:-4 5p}; do
  for v in ${VARIANTS:-prof cnt}; do
    [ -f basal_amd/lib/libbasal_amd_$v.so ] || continue
    BASAL_LIB=basal_amd/lib/libbasal_amd_$v.so python3 bench.py --config $cfg --genome realistic --steps 1 --warmup 1 --cpu-sample 0 --ref-sample 0 \
      > gpurun_out/${v}_c${cfg}.json 2> gpurun_out/${v}_c${cfg}.err
    echo "== $v config $cfg"; grep -E "^\[basal" gpurun_out/${v}_c${cfg}.err | tail -12
  done
done
