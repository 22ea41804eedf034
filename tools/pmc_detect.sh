#!/bin/bash
# Developer tool (GPU box): PMC counters for the pipeline kernels, one pass per counter group.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
B=${BATCH:-2048}
i=0
for GROUP in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
             "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" \
             "GRBM_GUI_ACTIVE GRBM_COUNT" ; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --kernel-trace --pmc $GROUP --output-format csv -d $OLDPWD/gpurun_out/pmc_$i -o pmc -- python3 $OLDPWD/tools/stage_times.py $B 1 > /dev/null 2>&1)
  python3 - "$PWD/gpurun_out/pmc_$i" <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[1].split("::")[-1] if "::" in r["Kernel_Name"] else r["Kernel_Name"][:30]
        d[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
for k in d:
    if k.startswith("k_synth"): continue
    print(k, {c: round(v / n[(k, c)]) for c, v in d[k].items()})
PY
done
