#!/bin/bash
# GPU box: kernel start / end times of one unprofiled pipeline step (do the two queues overlap?)
cd "$(dirname "$0")/../.."
ROOT=$PWD
export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/otrace; mkdir -p $ROOT/gpurun_out/otrace
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/otrace -o ot -- python3 $ROOT/tools/time_pipeline.py ${BATCH:-65536} 1 > /dev/null 2>&1)
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$ROOT/gpurun_out/otrace/**/ot_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("::")[-1].split("(")[0][:28], r.get("Queue_Id", "?")))
rows.sort()
t0 = rows[0][0]
for s, e, n, q in rows[-16:]:
    print("%-30s q=%s start %9.3f ms  end %9.3f ms  dur %7.3f" % (n, q, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
PY
