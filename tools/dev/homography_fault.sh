#!/bin/bash
# Developer tool (GPU box; builds from the tree as it is): the one queue pattern under which k_homography's transient fault
# of round 5 shows -- frame chunks of the whole pipeline in flight on four queues (capi.cpp -DDMZ_DEV_PIPE_CHUNKS,
# DMZ_HIP_PIPE_CHUNKS=8) -- with the kernel's self-check compiled OUT (-DDMZ_HOMOGRAPHY_NOCHECK) or IN, and tools/dev/determinism.py
# comparing every record and card of `reps` passes with the first.
# usage: tools/dev/homography_fault.sh [nocheck|check|check162|selfcheck|trace|wait|pad256|padn|canary] [reps] [chunks] [extra geometry.hip flags]
#   nocheck / wait / selfcheck: the 162-register kernel of round 5, single evaluation; pad256: single evaluation, 256 registers;
#   check: the shipped kernel (256 registers + self-check); check162: round 5's shipped kernel
#   trace: k_homography evaluated twice with a hash of its state after every Householder step; pairs that disagree are dumped
cd "$(dirname "$0")/../.."
MODE=${1:-nocheck}; REPS=${2:-12}; CHUNKS=${3:-8}; shift 3 2>/dev/null
P=card.io-dmz_amd; D=gpurun_out/hfault; mkdir -p $D
HF="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -I$P/csrc"
GFLAGS=""; WFLAGS=""
case $MODE in
  nocheck) GFLAGS="-DDMZ_HOMOGRAPHY_NOCHECK -DDMZ_HOMOGRAPHY_PAD256=0";;
  selfcheck) GFLAGS="-DDMZ_HOMOGRAPHY_NOCHECK -DDMZ_HOMOGRAPHY_PAD256=0 -DDMZ_DEV_SELFCHECK"; WFLAGS="-DDMZ_DEV_SELFCHECK";;
  trace) GFLAGS="-DDMZ_DEV_HTRACE -DDMZ_HOMOGRAPHY_PAD256=0";;  # 182 registers: still beside two 160-register waves
  wait) GFLAGS="-DDMZ_HOMOGRAPHY_NOCHECK -DDMZ_HOMOGRAPHY_PAD256=0 -DDMZ_DEV_HWAIT";;
  pad256) GFLAGS="-DDMZ_HOMOGRAPHY_NOCHECK";;
  check162) GFLAGS="-DDMZ_HOMOGRAPHY_PAD256=0";;
  canary) GFLAGS="-DDMZ_HOMOGRAPHY_NOCHECK -DDMZ_HOMOGRAPHY_PAD256=0 -DDMZ_DEV_HCANARY";;  # v164 .. v191 hold a pattern: 192 registers
  padn) GFLAGS="-DDMZ_HOMOGRAPHY_NOCHECK -DDMZ_HOMOGRAPHY_PAD256=0";;  # + extra flag -DDMZ_DEV_HPADN=<last register index>
esac
/opt/rocm/bin/hipcc $HF -DDMZ_DEV_PIPE_CHUNKS -x hip -c $P/csrc/capi.cpp -o $D/capi.o 2>/dev/null
/opt/rocm/bin/hipcc $HF $GFLAGS "$@" -c $P/csrc/geometry.hip -o $D/geometry.o 2>/dev/null
/opt/rocm/bin/hipcc $HF $WFLAGS -c $P/csrc/warp.hip -o $D/warp.o 2>/dev/null
OBJS=""
for f in detect vseg hseg digits expiry session plumbing synth weights_blob; do OBJS="$OBJS $P/csrc/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/lib_$MODE.so $OBJS $D/capi.o $D/geometry.o $D/warp.o -ldl
echo "== $MODE, $CHUNKS chunks, $REPS passes of ${BATCH:-65536} frames"
DMZ_HIP_LIB=$PWD/$D/lib_$MODE.so DMZ_HIP_PIPE_CHUNKS=$CHUNKS python tools/dev/determinism.py ${BATCH:-65536} $REPS 2>&1 | grep -v "amdgpu.ids" | tail -${TAIL:-12}
