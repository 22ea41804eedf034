#!/bin/bash
# Developer tool (GPU box): A/B the timed step of bench.py (three queues) between the committed (HEAD) version of one .hip
# file and the working copy, alternating on ONE box (boxes differ by +-1.5 %: differences below that need the same box).
# Needs tools/ab_prepare.sh <file.hip> first (no git on the GPU box).   usage: tools/ab_bench.sh <file.hip> [repetitions]
set -e
cd "$(dirname "$0")/.."
FILE=$1; REPS=${2:-3}
P=card.io-dmz_amd
mkdir -p gpurun_out/ab
# the Makefile's per-file flags
EXTRA_FILE_FLAGS=""; case $FILE in vseg.hip|expiry.hip) EXTRA_FILE_FLAGS="-fno-slp-vectorize";; esac
for V in head work; do
  SRC=$P/csrc/$FILE
  if [ $V == head ]; then cp $P/csrc/$FILE.head gpurun_out/ab/$FILE; SRC=gpurun_out/ab/$FILE; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -I$P/csrc $EXTRA_FILE_FLAGS \
     -c $SRC -o gpurun_out/ab/$V.o 2>/dev/null
  OBJS=""
  for f in detect geometry warp vseg hseg digits expiry session plumbing synth capi weights_blob; do
    if [ "$f.hip" == "$FILE" ]; then OBJS="$OBJS gpurun_out/ab/$V.o"; else OBJS="$OBJS $P/csrc/$f.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/ab/lib_$V.so $OBJS -ldl
done
if [ -n "$AB_STAGES" ]; then  # per-stage hipEvent times (one queue) instead of the timed step
  for rep in $(seq $REPS); do for V in head work; do echo -n "$V: "; DMZ_HIP_LIB=$PWD/gpurun_out/ab/lib_$V.so python tools/stage_times.py ${AB_BATCH:-65536} 2 2>/dev/null | cut -d" " -f3-; done; done
  exit 0
fi
for rep in $(seq $REPS); do
  for V in head work; do
    echo -n "$V: "
    DMZ_HIP_LIB=$PWD/gpurun_out/ab/lib_$V.so python bench.py --no-cpu-baseline 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms per step', d['value'], 'frames/s')"
  done
done
