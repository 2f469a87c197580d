#!/bin/bash
# Round-4 evidence for the lines not covered by tools/run_profile_r04_gap.sh: the headline (plain, under rocprofv3 --kernel-trace --stats, counter
# passes), config 3 (paired-end instantiations) and the uniform genome.   gpurun --timeout 1200 -- 'bash tools/run_profile_r04.sh r04'
TAG=${1:-r04}
ROOT=$(pwd)
export TMPDIR=/tmp
python3 bench.py > gpurun_out/${TAG}_bench_noprof.json 2> gpurun_out/${TAG}_bench_noprof.err
echo "[profile] plain bench done: $(cut -c80-140 gpurun_out/${TAG}_bench_noprof.json)"
cd /tmp
# (one placement under the profiler: every launch the trace sees is then on the placement the steps run on, and its mean is comparable with the HIP events)
BASAL_BENCH_PLACEMENT_DRAWS=1 BASAL_BENCH_NO_UNIFORM=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_$TAG" -- python3 "$ROOT/bench.py" > "$ROOT/gpurun_out/${TAG}_bench.json" 2> "$ROOT/gpurun_out/${TAG}_bench.err"
cd "$ROOT"
echo "[profile] kernel-trace bench done: $(cut -c80-140 gpurun_out/${TAG}_bench.json)"
f=$(ls -t gpurun_out/prof_$TAG/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${TAG}_bench_kernel_stats.csv && grep -h "align_kernel\|fill_flanks" gpurun_out/${TAG}_bench_kernel_stats.csv | cut -c1-60,150-400
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete
bash tools/run_pmc.sh $TAG "--cpu-sample 0 --ref-sample 0 --steps 3" 2>&1 | grep -v "^    @" | tail -26
export BASAL_BENCH_NO_UNIFORM=1
python3 bench.py --config 3 --steps 3 > gpurun_out/${TAG}_c3_bench.json 2> gpurun_out/${TAG}_c3_bench.err || true
echo "[profile] config 3: $(cut -c80-160 gpurun_out/${TAG}_c3_bench.json)"
bash tools/run_pmc.sh ${TAG}_c3 "--config 3 --cpu-sample 0 --ref-sample 0 --steps 3" 2>&1 | grep -v "^    @" | tail -26
python3 bench.py --genome uniform --steps 5 > gpurun_out/${TAG}_uniform_bench.json 2> gpurun_out/${TAG}_uniform_bench.err || true
echo "[profile] uniform genome: $(cut -c80-150 gpurun_out/${TAG}_uniform_bench.json)"
