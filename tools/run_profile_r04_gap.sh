#!/bin/bash
# Round 4: the workloads whose counter passes were missing (missing 4): configs 4 / 5p / 5 on the hg38-like genome and the 150-base
# kernels -- bench line (oracle-checked sample) + the rocprofv3 passes of tools/run_pmc.sh each.
#   gpurun --timeout 1200 -- 'bash tools/run_profile_r04_gap.sh r04'
TAG=${1:-r04}
export TMPDIR=/tmp
export BASAL_BENCH_NO_UNIFORM=1
for C in 4 5p 5; do
  python3 bench.py --config $C --genome realistic --steps 3 > gpurun_out/${TAG}_c${C}_realistic_bench.json 2> gpurun_out/${TAG}_c${C}_realistic_bench.err || true
  echo "[profile] config $C, hg38-like genome: $(cut -c80-150 gpurun_out/${TAG}_c${C}_realistic_bench.json)"
  bash tools/run_pmc.sh ${TAG}_c${C}_realistic "--genome realistic --config $C --cpu-sample 0 --ref-sample 0 --steps 2" 2>&1 | grep -v "^    @" | tail -24
done
python3 bench.py --read-len 150 --steps 3 > gpurun_out/${TAG}_150bp_realistic_bench.json 2> gpurun_out/${TAG}_150bp_realistic_bench.err || true
echo "[profile] 150 bp reads, hg38-like genome: $(cut -c80-150 gpurun_out/${TAG}_150bp_realistic_bench.json)"
bash tools/run_pmc.sh ${TAG}_150bp_realistic "--genome realistic --read-len 150 --cpu-sample 0 --ref-sample 0 --steps 2" 2>&1 | grep -v "^    @" | tail -24
python3 bench.py --read-len 150 --gap 2 --genome uniform --steps 3 > gpurun_out/${TAG}_150bp_g2_bench.json 2> gpurun_out/${TAG}_150bp_g2_bench.err || true
echo "[profile] 150 bp reads -g 2, uniform genome: $(cut -c80-150 gpurun_out/${TAG}_150bp_g2_bench.json)"
bash tools/run_pmc.sh ${TAG}_150bp_g2 "--genome uniform --read-len 150 --gap 2 --cpu-sample 0 --ref-sample 0 --steps 2" 2>&1 | grep -v "^    @" | tail -24
