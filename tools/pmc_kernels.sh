#!/bin/bash
# Developer tool (GPU box): per-kernel means of a few PMC counters over tools/stage_times.py
# usage: tools/pmc_kernels.sh COUNTER [COUNTER ...]   (one rocprofv3 pass per counter, kernel-trace only)
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
cd /tmp
for C in "$@"; do
  OUT=$ROOT/gpurun_out/pmck_$C
  rm -rf $OUT
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o pmc -- python3 $ROOT/tools/stage_times.py ${BATCH:-2048} 1 > /dev/null 2>&1
  python3 - "$OUT" $C <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != sys.argv[2]: continue
        name = r["Kernel_Name"]
        k = name.split("::")[-1].split("(")[0] if "::" in name else name[:40]
        d[k].append(float(r["Counter_Value"]))
print("# %s per dispatch" % sys.argv[2])
for k, v in sorted(d.items()):
    print("%-40s n=%d mean=%.4g" % (k, len(v), sum(v) / len(v)))
PY
  rm -rf $OUT
done
