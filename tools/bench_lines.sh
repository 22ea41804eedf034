#!/bin/bash
# GPU box: the round's bench.py lines once more, AFTER tools/profile_round.sh <tag>'s summaries were copied to profiles/ and
# the tag written to profiles/CURRENT -- so that roofline.traffic / counters_from of the committed lines name that profile's
# own counters (profile_round.sh runs bench.py before its PMC passes exist).   usage: tools/bench_lines.sh <tag>
TAG=$1
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
python3 bench.py > $OUT/${TAG}_bench_batch65536.json 2> $OUT/bench_lines.log
python3 bench.py --config 2 > $OUT/${TAG}_bench_config2_batch4096.json 2>> $OUT/bench_lines.log
python3 bench.py --config 3 > $OUT/${TAG}_bench_config3_batch65536.json 2>> $OUT/bench_lines.log
python3 bench.py --corpus mixed --no-cpu-baseline > $OUT/${TAG}_bench_mixed_batch65536.json 2>> $OUT/bench_lines.log
cut -c1-200 $OUT/${TAG}_bench_batch65536.json
