#!/bin/bash
# Developer tool: device assembly + register/LDS summary of one kernel file -> /tmp/<name>.s
# usage: tools/asm.sh warp [extra hipcc flags]
F=$1; shift
cd "$(dirname "$0")/../card.io-dmz_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
  -S --cuda-device-only -I../../include "$@" -o /tmp/$F.s $F.hip 2>&1 | grep -v hip-link
grep -n "^; Kernel\|NumVgprs\|NumAgprs\|ScratchSize\|Occupancy\|LDSByteSize\|^_Z.*:$" /tmp/$F.s | grep -v "^.*\.L" | head -60
