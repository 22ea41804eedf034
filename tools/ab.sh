#!/bin/bash
# Developer tool (GPU box): A/B the committed (HEAD) version of one .hip file against the working copy.
# Needs tools/ab_prepare.sh <file.hip> to have been run in the build container first (no git on the GPU box):
# it drops the HEAD version next to the working copy as csrc/<file>.head
set -e
cd "$(dirname "$0")/.."
FILE=$1
P=card.io-dmz_amd
mkdir -p gpurun_out/ab
for V in head work; do
  SRC=$P/csrc/$FILE
  if [ $V == head ]; then cp $P/csrc/$FILE.head gpurun_out/ab/$FILE; SRC=gpurun_out/ab/$FILE; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -I$P/csrc \
     -c $SRC -o gpurun_out/ab/$V.o 2>/dev/null
  OBJS=""
  for f in detect geometry warp vseg hseg digits expiry session plumbing synth capi weights_blob; do
    if [ "$f.hip" == "$FILE" ]; then OBJS="$OBJS gpurun_out/ab/$V.o"; else OBJS="$OBJS $P/csrc/$f.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/ab/lib_$V.so $OBJS
done
for rep in 1 2; do
  for V in head work; do
    echo -n "$V: "; DMZ_HIP_LIB=$PWD/gpurun_out/ab/lib_$V.so python tools/stage_times.py ${BATCH:-8192} 3 | sed 's/.*B=/B=/'
  done
done
