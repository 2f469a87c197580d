#!/bin/bash
# Regression sweep: every kernel family's bench line with two builds of the library (tools/run_ab_sweep.sh old.so new.so)
cd "$(dirname "$0")/.."
export BASAL_BENCH_NO_H2H=1 BASAL_BENCH_NO_UNIFORM=1
run() {  # label, bench arguments...
  local label=$1; shift
  for L in $LIBS; do
    BASAL_LIB=$L python3 bench.py "$@" --steps 3 --cpu-sample 100000 --ref-sample 0 2>gpurun_out/ab_sweep.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %-36s %8.2f %s  kernel %8.2f ms  %s' % ('$label', '$L', d['value'], d['unit'], d['roofline']['kernel_ms'], d['cpu_baseline']['sample'][-40:]))" || tail -3 gpurun_out/ab_sweep.err
  done
}
LIBS="$*"
mkdir -p gpurun_out
run "c2 realistic" --config 2
run "c2 uniform" --config 2 --genome uniform
run "c5 realistic" --config 5
run "c5 uniform" --config 5 --genome uniform
run "c4 uniform" --config 4 --genome uniform
run "c5p uniform" --config 5p --genome uniform
run "150bp realistic" --config 2 --read-len 150 --batch 4000000
run "150bp uniform" --config 2 --read-len 150 --genome uniform --batch 4000000
run "150bp g2 uniform" --config 2 --read-len 150 --gap 2 --genome uniform --batch 2000000
run "300bp uniform" --config 2 --read-len 300 --genome uniform --batch 2000000
run "300bp realistic" --config 2 --read-len 300 --batch 1000000
