#!/bin/bash
# GPU box: stage timings of every library under variants/ (built by tools/dev/ablate_local.sh)
cd "$(dirname "$0")/../.."
python tools/stage_times.py ${BATCH:-8192} 3
for L in variants/*.so; do
  echo "== $L"
  DMZ_HIP_LIB=$PWD/$L python tools/stage_times.py ${BATCH:-8192} 3
done
