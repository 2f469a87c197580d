#!/bin/bash
# A/B of libraries on one bench line: tools/run_ab_one.sh "<config> <genome>" lib...
cd "$(dirname "$0")/.."
export BASAL_BENCH_NO_H2H=1 BASAL_BENCH_NO_UNIFORM=1
set -- $1 "${@:2}"
cfg=$1; G=$2; shift 2
for L in "$@"; do
  BASAL_LIB=$L python3 bench.py --config $cfg --genome $G --steps 3 --cpu-sample 100000 --ref-sample 0 --placement-draws ${DRAWS:-1} 2>gpurun_out/ab_one.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L config $cfg $G: %.2f Mreads/s kernel %.2f ms  %s' % (d['value'], d['roofline']['kernel_ms'], d['cpu_baseline']['sample'][:60]))" || tail -3 gpurun_out/ab_one.err
done
