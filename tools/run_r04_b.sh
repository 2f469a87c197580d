#!/bin/bash
cd "$(dirname "$0")/.."
export BASAL_BENCH_NO_H2H=1
echo "== HEAVY=1 parity (fixtures)"
BASAL_HEAVY=1 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -x -m gpu -p no:cacheprovider -k "test_hit_logs_match_oracle or test_sam_matches_golden_through_abi or test_small_batches_carry_state" 2>&1 | tail -3
echo "== chk twin HEAVY=1 gap fixtures"
BASAL_LIB=basal_amd/lib/libbasal_amd_chk.so BASAL_HEAVY=1 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -x -m gpu -p no:cacheprovider -k "test_hit_logs_match_oracle and (g1 or g2 or g3 or pipeline or gact)" 2>&1 | tail -3
echo "== at-scale heavy gap configs"
timeout -k 10 600 python3 -m pytest tests/test_gpu_scale.py -q -x -m gpu -p no:cacheprovider -k "test_config_matches_oracle_on_sample_and_properties and (c4_realistic_heavy or c5p_realistic_heavy or ct_150_g2_realistic_heavy)" 2>&1 | tail -5
for cfg in 4 5p; do
  echo "== bench config $cfg realistic"
  timeout -k 10 400 python3 bench.py --config $cfg --genome realistic --steps 2 --warmup 1 --cpu-sample 200000 --ref-sample 0 > gpurun_out/r04b_c${cfg}_realistic.json 2> gpurun_out/r04b_c${cfg}_realistic.err || tail -5 gpurun_out/r04b_c${cfg}_realistic.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r04b_c${cfg}_realistic.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['cpu_baseline']['sample'][:80])
"
done
VARIANTS=prof bash tools/run_prof_gap.sh 2>&1 | grep -E "^==|basal phases\]|basal counts"
