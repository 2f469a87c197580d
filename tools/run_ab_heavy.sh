#!/bin/bash
# A/B of libraries on the HEAVY non-GAP lines: the headline (config 2, hg38-like genome) and config 5 on that genome; 200 000-read oracle samples
cd "$(dirname "$0")/.."
export BASAL_BENCH_NO_H2H=1 BASAL_BENCH_NO_UNIFORM=1
for L in "$@"; do
  for cfg in 2 5; do
    BASAL_LIB=$L python3 bench.py --config $cfg --steps 4 --cpu-sample 200000 --ref-sample 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L config $cfg: %.2f Mreads/s kernel %.2f ms  %s' % (d['value'], d['roofline']['kernel_ms'], d['cpu_baseline']['sample'][:50]))"
  done
done
