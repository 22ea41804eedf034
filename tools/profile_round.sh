#!/bin/bash
# Run on the GPU box (via gpurun): collects the round's profile summaries into gpurun_out/profiles_$1/
#   1. bench.py lines: config 4 (default command), config 2, config 3, and the mixed-corpus line (--corpus mixed)
#   2. rocprofv3 --kernel-trace --stats of the default bench.py command (minus the CPU baseline leg)
#   3. PMC passes (separate runs, kernel-trace only): FETCH_SIZE, WRITE_SIZE per kernel (batch 4096)
#   4. SQ counter passes (batch 8192): VALU / MFMA / LDS / wait counters per kernel
# Copy what should be judged into profiles/ and name the tag in profiles/CURRENT.
TAG=${1:-r2}
B=${BATCH:-65536}
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 $ROOT/bench.py > $OUT/${TAG}_bench_batch$B.json 2> $OUT/bench.log
python3 $ROOT/bench.py --config 2 > $OUT/${TAG}_bench_config2_batch4096.json 2>> $OUT/bench.log
python3 $ROOT/bench.py --config 3 > $OUT/${TAG}_bench_config3_batch65536.json 2>> $OUT/bench.log
python3 $ROOT/bench.py --corpus mixed --no-cpu-baseline > $OUT/${TAG}_bench_mixed_batch$B.json 2>> $OUT/bench.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $ROOT/bench.py --batch $B --steps 3 --warmup 1 --no-cpu-baseline --no-side-configs > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
cp $OUT/trace/bench_kernel_stats.csv $OUT/${TAG}_kernel_stats_batch$B.csv 2>/dev/null || find $OUT/trace -name '*kernel_stats.csv' -exec cp {} $OUT/${TAG}_kernel_stats_batch$B.csv \;
summarise() {  # dir, output file, batch, counters...
  python3 - "$@" <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
names = sys.argv[4:]
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        k = name.split("::")[-1].split("(")[0] if "::" in name else name[:40]
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[2], "w") as o:
    if len(names) == 1:
        o.write("# %s per dispatch (rocprofv3 units: KB as reported; FETCH_SIZE counts 128-byte requests at 64 bytes on gfx950: double it), tools/stage_times.py %s 1\n" % (names[0], sys.argv[3]))
        for k, v in sorted(d.items()):
            x = v[names[0]]
            if x:
                o.write("%-40s dispatches=%d mean=%.1f max=%.1f\n" % (k, len(x), sum(x) / len(x), max(x)))
    else:
        o.write("# per dispatch means, tools/stage_times.py %s 1 (SQ_* cycle counters count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles)\n" % sys.argv[3])
        o.write("%-34s %s\n" % ("kernel", " ".join("%24s" % n for n in names)))
        for k, v in sorted(d.items()):
            o.write("%-34s %s\n" % (k[:34], " ".join("%24.5g" % (sum(v[n]) / max(1, len(v[n]))) for n in names)))
PY
}
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -o pmc -- python3 $ROOT/tools/stage_times.py 4096 1 > /dev/null 2>&1
  summarise $OUT/pmc_$C $OUT/${TAG}_pmc_${C}_batch4096.txt 4096 $C
done
SQ1="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"
SQ2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS"
rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $OUT/pmc_sq1 -o pmc -- python3 $ROOT/tools/stage_times.py 8192 1 > /dev/null 2>&1
summarise $OUT/pmc_sq1 $OUT/${TAG}_pmc_SQ_issue_batch8192.txt 8192 $SQ1
rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $OUT/pmc_sq2 -o pmc -- python3 $ROOT/tools/stage_times.py 8192 1 > /dev/null 2>&1
summarise $OUT/pmc_sq2 $OUT/${TAG}_pmc_SQ_insts_batch8192.txt 8192 $SQ2
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_sq1 $OUT/pmc_sq2
ls -la $OUT
