#!/bin/bash
# Round-end evidence on the GPU box:  gpurun --timeout 1200 -- 'bash tools/run_profile.sh r03'
#   1. python3 bench.py (default flags: config 2 on the hg38-like genome) plain   -> gpurun_out/<tag>_bench_noprof.json
#   2. the same command under rocprofv3 --kernel-trace --stats                     -> gpurun_out/<tag>_bench.json + kernel stats
#   3. tools/run_pmc.sh (counter passes, --cpu-sample 0 --steps 3)                 -> gpurun_out/pmc_<tag>.csv
#   4. the other workloads: uniform genome, configs 4 / 5 / 5p (uniform genome, as rounds 1-2 measured them), config 3: bench line + counter passes
set -e
TAG=${1:-r03}
ROOT=$(pwd)
export TMPDIR=/tmp
python3 bench.py > gpurun_out/${TAG}_bench_noprof.json 2> gpurun_out/${TAG}_bench_noprof.err
echo "[profile] plain bench done: $(cut -c80-140 gpurun_out/${TAG}_bench_noprof.json)"
cd /tmp
BASAL_BENCH_NO_UNIFORM=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_$TAG" -- python3 "$ROOT/bench.py" > "$ROOT/gpurun_out/${TAG}_bench.json" 2> "$ROOT/gpurun_out/${TAG}_bench.err"
cd "$ROOT"
echo "[profile] kernel-trace bench done: $(cut -c80-140 gpurun_out/${TAG}_bench.json)"
grep -h "align_kernel\|fill_flanks\|emit_pairs\|split_counts" gpurun_out/prof_$TAG/*/*kernel_stats.csv | cut -c1-60,150-400 || true
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete
bash tools/run_pmc.sh $TAG "--cpu-sample 0 --ref-sample 0 --steps 3" 2>&1 | grep -v "^    @"
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
