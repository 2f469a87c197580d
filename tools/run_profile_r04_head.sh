#!/bin/bash
# the headline's plain line and the same command under rocprofv3 --kernel-trace --stats
TAG=${1:-r04}
ROOT=$(pwd)
export TMPDIR=/tmp
python3 bench.py > gpurun_out/${TAG}_bench_noprof.json 2> gpurun_out/${TAG}_bench_noprof.err
echo "[profile] plain bench done: $(cut -c80-140 gpurun_out/${TAG}_bench_noprof.json)"
cd /tmp
rm -rf "$ROOT/gpurun_out/prof_$TAG"
# (one placement under the profiler: every launch the trace sees is then on the placement the steps run on, and its mean is comparable with the HIP events)
BASAL_BENCH_PLACEMENT_DRAWS=1 BASAL_BENCH_NO_UNIFORM=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_$TAG" -- python3 "$ROOT/bench.py" > "$ROOT/gpurun_out/${TAG}_bench.json" 2> "$ROOT/gpurun_out/${TAG}_bench.err"
cd "$ROOT"
echo "[profile] kernel-trace bench done: $(cut -c80-140 gpurun_out/${TAG}_bench.json)"
f=$(ls -t gpurun_out/prof_$TAG/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${TAG}_bench_kernel_stats.csv && grep -h "align_kernel" gpurun_out/${TAG}_bench_kernel_stats.csv | cut -c1-400
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete
