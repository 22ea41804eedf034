#!/bin/bash
# GPU box: per-kernel PMC counters for every library under variants/ (built by tools/dev/ablate_local.sh).
# usage: [BATCH=2048] [COUNTERS="A B C"] [ONLY=k_digits] tools/dev/ablate_pmc.sh      (<= 8 SQ counters per pass)
cd "$(dirname "$0")/../.."
ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/ablate_pmc
rm -rf $OUT; mkdir -p $OUT
COUNTERS=${COUNTERS:-"SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VALU"}
run() {  # tag, lib
  cd /tmp
  DMZ_HIP_LIB=$2 rocprofv3 --kernel-trace --pmc $COUNTERS --output-format csv -d $OUT/$1 -o pmc -- python3 $ROOT/tools/stage_times.py ${BATCH:-2048} 1 > /dev/null 2>&1
  python3 - $OUT/$1 $1 "${ONLY:-k_}" ${BATCH:-2048} $COUNTERS <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
names = sys.argv[5:]
B = float(sys.argv[4])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        k = name.split("::")[-1].split("(")[0] if "::" in name else name[:40]
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(d.items()):
    if not k.startswith(sys.argv[3]) or k.startswith("k_synth"): continue
    m = {n: sum(x) / len(x) for n, x in v.items()}
    print("%-10s %-34s " % (sys.argv[2], k[:34]) + " ".join("%s %.0f" % (n.replace("SQ_", "").replace("INSTS_", "I_").replace("ACTIVE_INST_", "A_"), m.get(n, 0) / B) for n in names))
PY
  rm -rf $OUT/$1
}
echo "# per frame (counter / batch), batch ${BATCH:-2048}"
run default ""
for L in $ROOT/variants/*.so; do run $(basename $L .so) $L; done
