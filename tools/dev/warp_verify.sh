#!/bin/bash
# Developer tool (GPU box): k_warp compiled with -DDMZ_WARP_VERIFY compares EVERY cheap coordinate of the fast loops with the
# exact sequence (device printf on a mismatch) over the synthetic corpus: usage tools/dev/warp_verify.sh [frames] [seed]
set -e
cd "$(dirname "$0")/../.."
P=card.io-dmz_amd
mkdir -p gpurun_out/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -I$P/csrc -DDMZ_WARP_VERIFY \
   -c $P/csrc/warp.hip -o gpurun_out/ab/warp_verify.o 2>/dev/null
OBJS=""
for f in detect geometry warp vseg hseg digits expiry session plumbing synth capi weights_blob; do
  if [ $f == warp ]; then OBJS="$OBJS gpurun_out/ab/warp_verify.o"; else OBJS="$OBJS $P/csrc/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/ab/lib_warp_verify.so $OBJS -ldl
DMZ_HIP_LIB=$PWD/gpurun_out/ab/lib_warp_verify.so python - "$@" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
pkg = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
seed = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0xCA4D10
ctx = pkg.Context(0)
y = ctx.alloc(n * pkg.FRAME_BYTES); res = ctx.alloc(n * 1024); cards = ctx.alloc(n * pkg.CARD_BYTES)
ctx.synth_frames(seed, 0, n, y.ptr)
ctx.pipeline(y.ptr, n, res.ptr, cards.ptr)
ctx.synchronize()
print("verified %d frames x 115560 pixels (mismatches print above)" % n)
PY
