#!/bin/bash
# A/B of libraries on the GAP lines: configs 4 and 5p on the hg38-like genome (HEAVY GAP kernels) and on the uniform one (standard GAP kernels)
cd "$(dirname "$0")/.."
export BASAL_BENCH_NO_H2H=1 BASAL_BENCH_NO_UNIFORM=1
for G in realistic uniform; do
  for cfg in 4 5p; do
    for L in "$@"; do
      BASAL_LIB=$L python3 bench.py --config $cfg --genome $G --steps 3 --cpu-sample 100000 --ref-sample 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L config $cfg $G: %.2f Mreads/s kernel %.2f ms  %s' % (d['value'], d['roofline']['kernel_ms'], d['cpu_baseline']['sample'][:60]))"
    done
  done
done
