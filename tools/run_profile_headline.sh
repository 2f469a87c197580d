#!/bin/bash
# The headline part of tools/run_profile.sh alone (config 2 on the hg38-like genome): plain bench line, the same command under
# rocprofv3 --kernel-trace --stats, the counter passes; then the workloads that run the same HEAVY kernels (config 5 and 150 bp reads on that genome).
#   gpurun --timeout 1200 -- 'bash tools/run_profile_headline.sh r03'
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
cp "$(ls -t gpurun_out/prof_$TAG/*/*kernel_stats.csv | head -1)" gpurun_out/${TAG}_bench_kernel_stats.csv
grep -h "align_kernel\|fill_flanks" gpurun_out/${TAG}_bench_kernel_stats.csv | cut -c1-60,150-400 || true
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete
bash tools/run_pmc.sh $TAG "--cpu-sample 0 --ref-sample 0 --steps 3" 2>&1 | grep -v "^    @"
export BASAL_BENCH_NO_UNIFORM=1
python3 bench.py --config 5 --steps 3 > gpurun_out/${TAG}_c5_realistic_bench.json 2> gpurun_out/${TAG}_c5_realistic_bench.err || true
echo "[profile] config 5, hg38-like genome: $(cut -c80-150 gpurun_out/${TAG}_c5_realistic_bench.json)"
python3 bench.py --read-len 150 --steps 3 > gpurun_out/${TAG}_150bp_realistic_bench.json 2> gpurun_out/${TAG}_150bp_realistic_bench.err || true
echo "[profile] 150 bp reads, hg38-like genome: $(cut -c80-150 gpurun_out/${TAG}_150bp_realistic_bench.json)"
