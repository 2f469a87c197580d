#!/bin/bash
# Calibrates rocprofv3's HBM-traffic counters on the access shapes of the align kernel (tools/microbench_traffic.hip):
#   gpurun -- 'bash tools/run_calibration.sh r02'   -> gpurun_out/cal_<tag>.csv  (copy to profiles/)
# One pass per counter group (FETCH_SIZE alone; TCC_MISS/TCC_HIT; request counters), never combined with tracing.
set -e
TAG=${1:-cal}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/cal_$TAG
mkdir -p "$OUT" "$ROOT/tools/_bin"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "$ROOT/tools/microbench_traffic.hip" -o "$ROOT/tools/_bin/mbt"
export TMPDIR=/tmp
cd /tmp
"$ROOT/tools/_bin/mbt" 8 > "$OUT/known.csv"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- "$ROOT/tools/_bin/mbt" 8 > /dev/null 2> "$OUT/stats.err"
i=0
for g in "FETCH_SIZE" "TCC_MISS_sum TCC_HIT_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum"; do
  timeout -k 10 300 rocprofv3 --pmc $g --output-format csv -d "$OUT/pmc$i" -- "$ROOT/tools/_bin/mbt" 8 > /dev/null 2> "$OUT/pmc$i.err" || echo "[cal] group $i ($g) failed"
  i=$((i+1))
done
cd "$ROOT"
python3 - "$OUT" > "gpurun_out/cal_$TAG.csv" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
known = {}
for row in csv.DictReader(l for l in open(os.path.join(out, "known.csv")) if not l.startswith("#")):
    known[row["kernel"]] = (int(row["algorithmic_bytes"]), int(row["sectors64_bytes"]))
dur = {}
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Name"].split("(")[0]
        if n in known:
            dur[n] = float(row["AverageNs"])
cnt = {}
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"].split("(")[0]
        if n in known:
            cnt.setdefault(n, {}).setdefault(row["Counter_Name"], 0.0)
            cnt[n][row["Counter_Name"]] += float(row["Counter_Value"])
names = sorted({c for v in cnt.values() for c in v})
print("# tools/run_calibration.sh: rocprofv3 --pmc <group> -- tools/_bin/mbt 8 (8 GiB table, every byte read once; one pass per counter group)")
print("# FETCH_SIZE is in KiB; ratio = FETCH_SIZE bytes / algorithmic bytes; ratio64 = FETCH_SIZE bytes / bytes of the distinct 64-byte sectors touched")
print("kernel,algorithmic_bytes,sectors64_bytes,avg_ns,GB_per_s_algorithmic," + ",".join(names) + ",fetch_ratio_algorithmic,fetch_ratio_sectors64")
for k, (a, s64) in known.items():
    c = cnt.get(k, {})
    fb = c.get("FETCH_SIZE", 0.0) * 1024
    print("%s,%d,%d,%.0f,%.1f,%s,%.3f,%.3f" % (k, a, s64, dur.get(k, 0), a / dur[k] if k in dur else 0, ",".join("%.6g" % c.get(n, 0) for n in names), fb / a if a else 0, fb / s64 if s64 else 0))
PY
cat "gpurun_out/cal_$TAG.csv"
rm -rf "$OUT"/pmc*/ "$OUT"/stats
