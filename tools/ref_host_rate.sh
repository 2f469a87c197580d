#!/bin/bash
# The unmodified reference host on the core (oracle/_ref_gpu/basal) with -p N worker threads: GPU-side rate of its 50 000-read batches.
#   gpurun -- 'bash tools/ref_host_rate.sh 16'   -> kernel calls, summed kernel time, span first-start..last-end of the align kernels
set -e
P=${1:-16}
ROOT=$(pwd)
D=/dev/shm/refhost_$$
mkdir -p $D
python3 - <<PY
import sys, os
sys.path.insert(0, "$ROOT"); sys.path.insert(0, "$ROOT/tests"); sys.path.insert(0, "$ROOT/tools")
import test_gpu_cli_scale as T
T.make_files("$D", "C:T", 2_000_000, 100, seed=71, scale=0.02, n_rate=0.0, many_n_frac=0.0, lowq_frac=0.0, adapter_frac=0.0, lower_frac=0.0)
PY
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $D/prof -- $ROOT/oracle/_ref_gpu/basal -a $D/r.fq -d $D/g.fa -M C:T -S 1 -s 12 -p $P -o /dev/null > $D/log 2>&1
tail -3 $D/log
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$D/prof/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "align_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
busy = sum(e - s for s, e in rows)
# union of the intervals (kernels of different lanes overlap)
u, cur_s, cur_e = 0, None, None
for s, e in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None: u += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
if cur_e is not None: u += cur_e - cur_s
span = rows[-1][1] - rows[0][0]
n = 2_000_000
print("align kernels: %d launches, sum %.2f ms, union (GPU busy with align) %.2f ms, span %.2f ms" % (len(rows), busy / 1e6, u / 1e6, span / 1e6))
print("GPU-side rate = reads / union = %.1f Mreads/s; host-paced rate = reads / span = %.1f Mreads/s" % (n / (u / 1e9) / 1e6, n / (span / 1e9) / 1e6))
PY
rm -rf $D
