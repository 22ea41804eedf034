#!/bin/bash
# Developer tool (GPU box or build container): gpurun_out/ab/lib_<name>.so = the library with ONE file recompiled with extra
# flags.  usage: tools/dev/variant_lib.sh <name> <file.hip> <extra hipcc flags...>
set -e
cd "$(dirname "$0")/../.."
NAME=$1; FILE=$2; shift 2
P=card.io-dmz_amd
D=${DMZ_AB_DIR:-gpurun_out/ab}   # (gpurun_out/ does not travel to the GPU box: build-container variants go to ab_libs/)
mkdir -p $D
EX=""; case $FILE in vseg.hip|expiry.hip) EX="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -I$P/csrc $EX "$@" \
   -c $P/csrc/$FILE -o $D/$NAME.o 2>/dev/null
OBJS=""
for f in detect geometry warp vseg hseg digits expiry session plumbing synth capi weights_blob; do
  if [ "$f.hip" == "$FILE" ]; then OBJS="$OBJS $D/$NAME.o"; else OBJS="$OBJS $P/csrc/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/lib_$NAME.so $OBJS -ldl
echo $D/lib_$NAME.so
