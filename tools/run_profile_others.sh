#!/bin/bash
# The part of tools/run_profile.sh behind the headline: uniform genome, configs 4 / 5 / 5p (uniform genome, as rounds 1-2 measured them), config 3 --
# bench line + counter passes each.   gpurun --timeout 1200 -- 'bash tools/run_profile_others.sh r03'
set -e
TAG=${1:-r03}
export TMPDIR=/tmp
export BASAL_BENCH_NO_UNIFORM=1
python3 bench.py --genome uniform --steps 5 > gpurun_out/${TAG}_uniform_bench.json 2> gpurun_out/${TAG}_uniform_bench.err || true
echo "[profile] uniform genome: $(cut -c80-150 gpurun_out/${TAG}_uniform_bench.json)"
bash tools/run_pmc.sh ${TAG}_uniform "--genome uniform --cpu-sample 0 --ref-sample 0 --steps 4" 2>&1 | grep -v "^    @" | tail -30
for C in 4 5 5p; do
  python3 bench.py --genome uniform --config $C --steps 5 > gpurun_out/${TAG}_c${C}_bench.json 2> gpurun_out/${TAG}_c${C}_bench.err || true
  echo "[profile] config $C: $(cut -c80-150 gpurun_out/${TAG}_c${C}_bench.json)"
  bash tools/run_pmc.sh ${TAG}_c${C} "--genome uniform --config $C --cpu-sample 0 --ref-sample 0 --steps 3" 2>&1 | grep -v "^    @" | tail -30
done
python3 bench.py --config 3 --steps 3 > gpurun_out/${TAG}_c3_bench.json 2> gpurun_out/${TAG}_c3_bench.err || true
echo "[profile] config 3: $(cut -c80-150 gpurun_out/${TAG}_c3_bench.json)"
bash tools/run_pmc.sh ${TAG}_c3 "--config 3 --cpu-sample 0 --ref-sample 0 --steps 3" 2>&1 | grep -v "^    @" | tail -30
