#!/bin/bash
# Developer tool (GPU box or build container): gpurun_out/ab/lib_<name>.so = the library with ONE file recompiled with extra
# flags.  usage: tools/dev/variant_lib.sh <name> <file.hip> <extra hipcc flags...>
set -e
cd "$(dirname "$0")/../.."
NAME=$1; FILE=$2; shift 2
P=card.io-dmz_amd
mkdir -p gpurun_out/ab
EX=""; case $FILE in vseg.hip|expiry.hip) EX="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -I$P/csrc $EX "$@" \
   -c $P/csrc/$FILE -o gpurun_out/ab/$NAME.o 2>/dev/null
OBJS=""
for f in detect geometry warp vseg hseg digits expiry session plumbing synth capi weights_blob; do
  if [ "$f.hip" == "$FILE" ]; then OBJS="$OBJS gpurun_out/ab/$NAME.o"; else OBJS="$OBJS $P/csrc/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/ab/lib_$NAME.so $OBJS -ldl
echo gpurun_out/ab/lib_$NAME.so
