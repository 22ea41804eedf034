#!/bin/bash
# GPU box: build one .hip file with extra flags and run a python script against that library
# usage: tools/dev/variant.sh <file.hip> "<flags>" <script.py>
cd "$(dirname "$0")/../.."
P=card.io-dmz_amd
FILE=$1; FLAGS=$2; SCRIPT=$3
mkdir -p gpurun_out/variant
# the Makefile's per-file flags
EXTRA_FILE_FLAGS=""; case $FILE in vseg.hip|expiry.hip) EXTRA_FILE_FLAGS="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude $EXTRA_FILE_FLAGS $FLAGS -c $P/csrc/$FILE -o gpurun_out/variant/x.o 2>/dev/null
OBJS=""
for f in detect geometry warp vseg hseg digits expiry session plumbing synth capi weights_blob; do
  if [ "$f.hip" == "$FILE" ]; then OBJS="$OBJS gpurun_out/variant/x.o"; else OBJS="$OBJS $P/csrc/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/variant/lib.so $OBJS
echo "== $FLAGS"
DMZ_HIP_LIB=$PWD/gpurun_out/variant/lib.so python $SCRIPT
