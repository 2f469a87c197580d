"""Summarise the rocprofv3 passes tools/run_pmc.sh wrote: per-launch means of every counter for the
align kernel (summed over the counter's instances, as rocprofv3 reports them), the per-read figure, and
the kernel's average duration from the --kernel-trace --stats pass.  Output: CSV on stdout."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
reads = 1_000_000
for j in glob.glob(os.path.join(out, "*.json")):
    try:
        d = json.loads(open(j).read().strip().splitlines()[-1])
        reads = d["config"].get("reads_per_step_per_gpu") or d["config"]["pairs_per_step"]  # (config 3: per read pair)
        break
    except Exception:
        pass
print("# tools/run_pmc.sh: rocprofv3 passes over `python3 bench.py --cpu-sample 0 --steps 4` (one pass per counter group, none combined with tracing)")
print("# per launch of the align kernel (%d reads), mean over the launches of the pass; FETCH_SIZE/WRITE_SIZE in KiB" % reads)
print("counter,mean_per_launch,per_read,launches")
for f in sorted(glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        if "align_kernel" in row["Name"]:
            print("kernel_avg_ns[%s],%s,%.4f,%s" % (row["Name"].split("::")[-1].split("(")[0], row["AverageNs"], float(row["AverageNs"]) / reads, row["Calls"]))
acc = {}
for f in sorted(glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    per = {}
    for row in csv.DictReader(open(f)):
        if "align_kernel" not in row["Kernel_Name"]:
            continue
        key = (row["Counter_Name"], row["Dispatch_Id"])
        per[key] = per.get(key, 0.0) + float(row["Counter_Value"])
    for (name, _), v in per.items():
        acc.setdefault(name, []).append(v)
for name in sorted(acc):
    v = acc[name]
    m = sum(v) / len(v)
    print("%s,%.6g,%.4g,%d" % (name, m, m / reads, len(v)))
