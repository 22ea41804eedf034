#!/bin/bash
# Developer tool (GPU box): rocprofv3 per-kernel stats of tools/stage_times.py
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p $ROOT/gpurun_out/kstats
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/kstats -o ks -- python3 $ROOT/tools/stage_times.py ${BATCH:-8192} 3 > /dev/null 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$ROOT/gpurun_out/kstats/**/ks_kernel_stats.csv", recursive=True) + glob.glob("$ROOT/gpurun_out/kstats/ks_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("::")[-1].split("(")[0][:40]
        print("%-42s calls=%s avg_us=%.1f pct=%s" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
    break
PY
