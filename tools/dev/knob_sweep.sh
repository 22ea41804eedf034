#!/bin/bash
# Developer tool (GPU box): one-knob-at-a-time sweep of the kernels' residency / batching switches; for each setting the
# stage's ms at 65 536 frames (two runs), beside the unchanged build's.  usage: tools/dev/knob_sweep.sh [detect|warp|scan|all]
cd "$(dirname "$0")/../.."
WHAT=${1:-all}
run() {  # name file stage flags...
  local name=$1 file=$2 stage=$3; shift 3
  if tools/dev/variant_lib.sh $name $file "$@" > /dev/null 2>&1; then
    for rep in 1 2; do
      echo -n "$stage $name ($*): "; DMZ_HIP_LIB=$PWD/gpurun_out/ab/lib_$name.so timeout 120 python tools/stage_times.py 65536 2 2>/dev/null | grep -o "$stage [0-9.]*" || echo failed
    done
  else echo "$name: build failed"; fi
  rm -f gpurun_out/ab/lib_$name.so gpurun_out/ab/$name.o
}
if [ $WHAT == detect ] || [ $WHAT == all ]; then
run base detect.hip detect
for w in 6 8; do run wpsH$w detect.hip detect -DDMZ_DETECT_WPS_H=$w; done
for w in 6 8; do run wpsV$w detect.hip detect -DDMZ_DETECT_WPS_V=$w; done
for f in 6 10 12; do run infl$f detect.hip detect -DDMZ_DT_INFLIGHT=$f; done
run base2 detect.hip detect
fi
if [ $WHAT == warp ] || [ $WHAT == all ]; then
run base warp.hip warp
for h in 128 144 160; do run lh$h warp.hip warp -DDMZ_WARP_LH=$h; done
for w in 2 4; do run waves$w warp.hip warp -DDMZ_WARP_WAVES=$w; done
run base2 warp.hip warp
fi
if [ $WHAT == scan ] || [ $WHAT == all ]; then
run base vseg.hip vseg
for b in 6 8; do run vblk$b vseg.hip vseg -DDMZ_VSEG_BLOCKS=$b; done
for b in 2 8; do run vscan$b vseg.hip vseg -DVS_SCAN_BATCH=$b; done
run base digits.hip digits
for b in 3 5; do run dwg$b digits.hip digits -DDMZ_DIGITS_WGS=$b; done
run base expiry.hip expiry_seg
for b in 2 4; do run xw$b expiry.hip expiry_seg -DDMZ_XSEG_WAVES=$b; done
run base hseg.hip hseg
for b in 3 5 6; do run hw$b hseg.hip hseg -DDMZ_HSEG_WAVES=$b; done
fi
