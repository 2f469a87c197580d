#!/bin/bash
# the four GAP bench lines (configs 4 / 5p on the hg38-like and the uniform genome), oracle-checked 200 000-read sample each
cd "$(dirname "$0")/.."
export BASAL_BENCH_NO_H2H=1
TAG=${1:-r04x}
for spec in ${SPECS:-"4:realistic" "5p:realistic" "4:uniform" "5p:uniform"}; do
  c=${spec%%:*}; g=${spec##*:}
  timeout -k 10 400 python3 bench.py --config $c --genome $g --steps 3 --warmup 1 --cpu-sample 200000 --ref-sample 0 > gpurun_out/${TAG}_c${c}_$g.json 2> gpurun_out/${TAG}_c${c}_$g.err || tail -5 gpurun_out/${TAG}_c${c}_$g.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/${TAG}_c${c}_$g.json').read().strip().splitlines()[-1])
print('config $c $g: %.2f Mreads/s  %s  %.1f ms  frac %.3f  %s' % (d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['cpu_baseline']['sample'][:60]))
"
done
