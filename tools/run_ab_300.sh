#!/bin/bash
# A/B of libraries on the 300-base lines (align_kernel<16,...>), uniform genome, plain and -g 1
cd "$(dirname "$0")/.."
export BASAL_BENCH_NO_H2H=1 BASAL_BENCH_NO_UNIFORM=1
mkdir -p gpurun_out
for G in 0 1; do
for L in "$@"; do
  BASAL_LIB=$L python3 bench.py --config 2 --read-len 300 --gap $G --genome uniform --batch 2000000 --steps 3 --cpu-sample 50000 --ref-sample 0 2>gpurun_out/ab_300.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L 300bp g$G uniform: %.2f Mreads/s kernel %.2f ms %s grid %s  %s' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel'], d['config'].get('kernel_grid'), d['cpu_baseline']['sample'][-40:]))" || tail -3 gpurun_out/ab_300.err
done
done
