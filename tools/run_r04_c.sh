#!/bin/bash
cd "$(dirname "$0")/.."
export BASAL_BENCH_NO_H2H=1
echo "== parity (fixtures), standard and HEAVY kernels; chk twin on the gap fixtures both ways"
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -x -m gpu -p no:cacheprovider -k "test_hit_logs_match_oracle or test_sam_matches_golden_through_abi" 2>&1 | tail -2
BASAL_HEAVY=1 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -x -m gpu -p no:cacheprovider -k "test_hit_logs_match_oracle or test_sam_matches_golden_through_abi or test_small_batches_carry_state" 2>&1 | tail -2
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -x -m gpu -p no:cacheprovider -k "test_gap_stream_bounds" 2>&1 | tail -2
echo "== at-scale gap configs (+ chk at scale)"
timeout -k 10 900 python3 -m pytest tests/test_gpu_scale.py -q -x -m gpu -p no:cacheprovider -k "(test_config_matches_oracle_on_sample_and_properties and (g2 or g1 or c4 or pipeline or c5p)) or test_gap_bounds_check_build_at_scale" 2>&1 | tail -3
for spec in "4 realistic" "5p realistic" "4 uniform" "5p uniform"; do
  set -- $spec
  echo "== bench config $1 $2"
  timeout -k 10 400 python3 bench.py --config $1 --genome $2 --steps 3 --warmup 1 --cpu-sample 200000 --ref-sample 0 > gpurun_out/r04c_c$1_$2.json 2> gpurun_out/r04c_c$1_$2.err || tail -5 gpurun_out/r04c_c$1_$2.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r04c_c$1_$2.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['cpu_baseline']['sample'][:80])
"
done
