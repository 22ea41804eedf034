#!/bin/bash
# GPU box: SQ_INSTS_VALU / SQ_INSTS_LDS / SQ_BUSY_CU_CYCLES per kernel for every library under variants/
cd "$(dirname "$0")/../.."
ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/ablate_pmc
rm -rf $OUT; mkdir -p $OUT
run() {  # tag, lib
  cd /tmp
  DMZ_HIP_LIB=$2 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_INSTS_SALU --output-format csv -d $OUT/$1 -o pmc -- python3 $ROOT/tools/stage_times.py ${BATCH:-2048} 1 > /dev/null 2>&1
  python3 - $OUT/$1 $1 <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        k = name.split("::")[-1].split("(")[0] if "::" in name else name[:40]
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(d.items()):
    if not k.startswith("k_") or k.startswith("k_synth"): continue
    m = {n: sum(x) / len(x) for n, x in v.items()}
    print("%-8s %-36s VALU %10.0f LDS %9.0f SALU %9.0f BUSY %10.0f" % (sys.argv[2], k[:36], m.get("SQ_INSTS_VALU", 0), m.get("SQ_INSTS_LDS", 0), m.get("SQ_INSTS_SALU", 0), m.get("SQ_BUSY_CU_CYCLES", 0)))
PY
  rm -rf $OUT/$1
}
run default ""
for L in $ROOT/variants/*.so; do run $(basename $L .so) $L; done
