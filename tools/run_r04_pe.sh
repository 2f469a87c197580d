#!/bin/bash
# paired-end parity (goldens through every form, at-scale vs the reference binary) + config 3 and the uniform-genome line
cd "$(dirname "$0")/.."
export BASAL_BENCH_NO_H2H=1 BASAL_BENCH_NO_UNIFORM=1
timeout -k 10 600 python3 -m pytest tests/ -q -x -m gpu -p no:cacheprovider -k "pe or pair or transcriptome or allmodes or paired" 2>&1 | tail -3
python3 bench.py --config 3 --steps 3 --warmup 1 > gpurun_out/${1:-r04g}_c3.json 2> gpurun_out/${1:-r04g}_c3.err || tail -5 gpurun_out/${1:-r04g}_c3.err
python3 -c "
import json
d=json.loads(open('gpurun_out/${1:-r04g}_c3.json').read().strip().splitlines()[-1])
c=d['config']; print('config 3: %.2f Mpairs/s host to host; kernels %.2f Mpairs/s (align %.2f ms + pair %.2f ms per %d pairs); paired %.4f; %s' % (d['value'], c['mpairs_per_s_kernels'], c['align_kernel_ms'], c['pair_kernel_ms'], c['pairs_per_step'], c['paired_frac'], (d['cpu_baseline'] or {}).get('sample','')[:70]))
"
python3 bench.py --genome uniform --steps 5 --cpu-sample 200000 --ref-sample 0 > gpurun_out/${1:-r04g}_uniform.json 2> gpurun_out/${1:-r04g}_uniform.err || tail -5 gpurun_out/${1:-r04g}_uniform.err
python3 -c "
import json
d=json.loads(open('gpurun_out/${1:-r04g}_uniform.json').read().strip().splitlines()[-1])
print('uniform genome: %.2f Mreads/s  %s  %.2f ms  frac %.3f' % (d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac']))
"
