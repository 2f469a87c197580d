#!/bin/bash
# round 4, first measurement of the HEAVY GAP kernels + LDS-poisoned check build on every fixture
cd "$(dirname "$0")/.."
export BASAL_BENCH_NO_H2H=1
for pz in 0xff 0xa5; do for hv in 0 1; do
  echo "== chk twin, BASAL_POISON=$pz BASAL_HEAVY=$hv"
  BASAL_LIB=basal_amd/lib/libbasal_amd_chk.so BASAL_POISON=$pz BASAL_HEAVY=$hv timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -x -m gpu -p no:cacheprovider \
    -k "test_hit_logs_match_oracle or test_sam_matches_golden_through_abi" 2>&1 | tail -4
done; done
for cfg in 4 5p; do
  echo "== bench config $cfg realistic"
  timeout -k 10 400 python3 bench.py --config $cfg --genome realistic --steps 2 --warmup 1 --cpu-sample 200000 --ref-sample 0 > gpurun_out/r04a_c${cfg}_realistic.json 2> gpurun_out/r04a_c${cfg}_realistic.err || tail -5 gpurun_out/r04a_c${cfg}_realistic.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r04a_c${cfg}_realistic.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['cpu_baseline']['sample'][:80])
"
done
