#!/bin/bash
# GPU box: one parity sweep of the whole pipeline against the oracle (tests/test_gpu_parity_large.py at sweep size).
# usage: tools/dev/sweep.sh <corpus frames> <fuzz frames> <seed> [flavour: 1 = the SSE2 order of the homography on both sides]
cd "$(dirname "$0")/../.."
DMZ_PARITY_FRAMES=$1 DMZ_FUZZ_FRAMES=$2 DMZ_PARITY_SEED=$3 DMZ_PARITY_FLAVOUR=${4:-0} timeout 5000 python -m pytest tests/test_gpu_parity_large.py -q -s -m gpu 2>&1 | \
  grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12
