#!/bin/bash
# quick A/B lines: config 3, the uniform genome, the headline (200 000-read oracle samples)
cd "$(dirname "$0")/.."
export BASAL_BENCH_NO_H2H=1 BASAL_BENCH_NO_UNIFORM=1
python3 bench.py --config 3 --steps 4 --warmup 1 --cpu-sample 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('config 3: %.2f h2h, kernels %.2f Mpairs/s, align %.2f ms' % (d['value'], c['mpairs_per_s_kernels'], c['align_kernel_ms']))"
python3 bench.py --genome uniform --steps 5 --cpu-sample 200000 --ref-sample 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('uniform: %.2f Mreads/s kernel %.2f ms' % (d['value'], d['roofline']['kernel_ms']))"
python3 bench.py --steps 6 --cpu-sample 200000 --ref-sample 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline: %.2f Mreads/s kernel %.2f ms' % (d['value'], d['roofline']['kernel_ms']))"
