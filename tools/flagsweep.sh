#!/bin/bash
# Developer tool (GPU box): rebuild every kernel file with extra compiler flags and print stage timings.
# usage: tools/flagsweep.sh "<flags A>" "<flags B>" ...      ("" = the Makefile's flags)
cd "$(dirname "$0")/.."
P=card.io-dmz_amd
mkdir -p gpurun_out/flags
i=0
for FLAGS in "$@"; do
  i=$((i+1))
  OBJS=""; ok=1
  for f in detect geometry warp vseg hseg digits expiry session plumbing synth; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude $FLAGS \
      -c $P/csrc/$f.hip -o gpurun_out/flags/${f}_$i.o 2> gpurun_out/flags/err_$i.txt || { ok=0; break; }
    OBJS="$OBJS gpurun_out/flags/${f}_$i.o"
  done
  echo "== [$FLAGS]"
  if [ $ok == 0 ]; then head -3 gpurun_out/flags/err_$i.txt; continue; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/flags/lib_$i.so $OBJS $P/csrc/capi.o $P/csrc/weights_blob.o
  for rep in 1 2; do DMZ_HIP_LIB=$PWD/gpurun_out/flags/lib_$i.so python tools/stage_times.py ${BATCH:-8192} 3 | sed 's/.*B=/B=/'; done
done
rm -f gpurun_out/flags/*.o
