#!/bin/bash
# Developer tool (GPU box): per-kernel means of a few PMC counters over tools/stage_times.py
# usage: tools/pmc_kernels.sh COUNTER [COUNTER ...]   (one rocprofv3 pass per counter, kernel-trace only)
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
cd /tmp
# an argument may group counters for one pass: A+B+C
for G in "$@"; do
  OUT=$ROOT/gpurun_out/pmck_tmp
  rm -rf $OUT
  rocprofv3 --kernel-trace --pmc ${G//+/ } --output-format csv -d $OUT -o pmc -- python3 $ROOT/tools/stage_times.py ${BATCH:-2048} 1 > /dev/null 2>&1
  python3 - "$OUT" $G <<'PY'
import csv, glob, sys, collections
names = sys.argv[2].split("+")
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        k = name.split("::")[-1].split("(")[0] if "::" in name else name[:40]
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# per dispatch: %-28s %s" % ("kernel", " ".join("%22s" % n for n in names)))
for k, v in sorted(d.items()):
    print("%-45s %s" % (k[:45], " ".join("%22.4g" % (sum(v[n]) / max(1, len(v[n]))) for n in names)))
PY
  rm -rf $OUT
done
