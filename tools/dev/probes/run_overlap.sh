#!/bin/bash
# GPU box: build and run the matrix / VALU overlap probe
cd "$(dirname "$0")"; mkdir -p ../../../gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu_overlap mfma_valu_overlap.hip 2>/dev/null && timeout 300 /tmp/mfma_valu_overlap
