#!/bin/bash
# Developer tool (run on the GPU box via gpurun): build variants of one .hip file with an
# extra -D flag each and print per-stage timings.  usage: tools/ablate.sh <file.hip> "<-Dflag>" ...
set -e
cd "$(dirname "$0")/.."
FILE=$1; shift
mkdir -p gpurun_out/ablate
P=card.io-dmz_amd
i=0
for FLAG in "$@"; do
  i=$((i+1))
  OUT=gpurun_out/ablate/lib_$i.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude $FLAG \
     -c $P/csrc/$FILE -o gpurun_out/ablate/var_$i.o 2>/dev/null
  OBJS=""
  for f in detect geometry warp vseg hseg digits expiry session plumbing synth capi weights_blob; do
    if [ "$f.hip" == "$FILE" ]; then OBJS="$OBJS gpurun_out/ablate/var_$i.o"; else OBJS="$OBJS $P/csrc/$f.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS
  echo "== $FLAG"
  DMZ_HIP_LIB=$PWD/$OUT python tools/stage_times.py ${BATCH:-8192} 3
done
