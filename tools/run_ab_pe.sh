#!/bin/bash
# A/B of libraries on the paired-end lines: config 3 (bench.py) and the command line on 4 M 100-base pairs (hg38-sized uniform genome)
cd "$(dirname "$0")/.."
export BASAL_BENCH_NO_H2H=1
for L in "$@"; do
  BASAL_LIB=$L python3 bench.py --config 3 --steps 4 --warmup 1 --cpu-sample 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$L config 3: %.2f h2h, kernels %.2f Mpairs/s, align %.2f ms' % (d['value'], c['mpairs_per_s_kernels'], c['align_kernel_ms']))"
done
