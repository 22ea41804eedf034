#!/bin/bash
# Run on the GPU box (via gpurun): collects the round's profile summaries into gpurun_out/profiles_$1/
#   1. rocprofv3 --kernel-trace --stats of bench.py (the bench.py default command line minus the CPU baseline)
#   2. PMC passes (separate runs, kernel-trace only): FETCH_SIZE, WRITE_SIZE per kernel
TAG=${1:-r1}
B=${BATCH:-65536}
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $ROOT/bench.py --batch $B --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
cp $OUT/trace/bench_kernel_stats.csv $OUT/${TAG}_kernel_stats_batch$B.csv 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -o pmc -- python3 $ROOT/tools/stage_times.py 4096 1 > /dev/null 2>&1
  python3 - "$OUT/pmc_$C" $C > $OUT/${TAG}_pmc_${C}_batch4096.txt <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != sys.argv[2]: continue
        name = r["Kernel_Name"]
        k = name.split("::")[-1].split("(")[0] if "::" in name else name[:40]
        d[k].append(float(r["Counter_Value"]))
print("# %s per dispatch (rocprofv3 units: KB as reported; see MI355X_MICROARCH.md HBM section for the gfx950 correction)" % sys.argv[2])
for k, v in sorted(d.items()):
    print("%-40s dispatches=%d mean=%.1f max=%.1f" % (k, len(v), sum(v) / len(v), max(v)))
PY
done
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
ls -la $OUT
