#!/bin/bash
# Build ablation variants HERE (hipcc cross-compiles) into variants/ (git-ignored *.so, travels with gpurun),
# then on the GPU box: tools/dev/ablate_run.sh.   usage: tools/dev/ablate_local.sh <file.hip> <tag> "<flags>" [<tag> "<flags>" ...]
cd "$(dirname "$0")/../.."
P=card.io-dmz_amd
FILE=$1; shift
mkdir -p variants
while [ $# -gt 1 ]; do
  TAG=$1; FLAGS=$2; shift 2
  (
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -I$P/csrc $FLAGS -c $P/csrc/$FILE -o variants/$TAG.o 2>/dev/null
  OBJS=""
  for f in detect geometry warp vseg hseg digits expiry session plumbing synth capi weights_blob; do
    if [ "$f.hip" == "$FILE" ]; then OBJS="$OBJS variants/$TAG.o"; else OBJS="$OBJS $P/csrc/$f.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/$TAG.so $OBJS && rm variants/$TAG.o && echo built $TAG
  ) &
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 1; done
done
wait
