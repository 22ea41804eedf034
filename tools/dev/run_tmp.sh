#!/bin/bash
cd "$(dirname "$0")/../.."
python -m pytest tests -m gpu -x -q 2>&1 | grep "passed\|failed\|Error\|assert" | head
for i in 1 2 3 4; do
  for V in head work; do
    if [ $V == work ]; then L=$PWD/card.io-dmz_amd/libdmz_hip.so; else L=$PWD/ab_libs/lib_$V.so; fi
    echo -n "$V: "; DMZ_HIP_LIB=$L python tools/stage_times.py 65536 2 2>/dev/null | cut -d" " -f9-10
  done
done
for i in 1 2; do for V in head work; do
  if [ $V == work ]; then L=$PWD/card.io-dmz_amd/libdmz_hip.so; else L=$PWD/ab_libs/lib_$V.so; fi
  echo -n "$V: "; DMZ_HIP_LIB=$L python tools/dev/pipe_ab.py 65536 8
done; done
bash tools/dev/sweep.sh 32768 16800 2468 2>&1 | tail -5
