#!/bin/bash
# GPU box: rocprofv3 kernel stats (detect kernels) for the default library and every library under variants/
cd "$(dirname "$0")/../.."
ROOT=$PWD
export TMPDIR=/tmp
K=${KERNELS:-k_detect}
run() {
  rm -rf $ROOT/gpurun_out/kstats; mkdir -p $ROOT/gpurun_out/kstats
  (cd /tmp && DMZ_HIP_LIB=$2 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/kstats -o ks -- python3 $ROOT/tools/stage_times.py ${BATCH:-8192} 3 > /dev/null 2>&1)
  python3 - $1 $K <<PY
import csv, glob, sys
for f in glob.glob("$ROOT/gpurun_out/kstats/**/ks_kernel_stats.csv", recursive=True) + glob.glob("$ROOT/gpurun_out/kstats/ks_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("::")[-1].split("(")[0][:44]
        if sys.argv[2] in n: print("%-8s %-46s avg_us=%.1f" % (sys.argv[1], n, float(r["AverageNs"]) / 1e3))
    break
PY
}
run default ""
for L in $ROOT/variants/*.so; do run $(basename $L .so) $L; done
