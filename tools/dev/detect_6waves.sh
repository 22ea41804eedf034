#!/bin/bash
# GPU box, timing only (wrong edges): k_detect_walk<hz> on top / bottom boxes narrowed to 372 columns = six 62-column waves instead
# of seven (the seventh wave of the real 389-column box owns 17 columns).  Bounds what ONE workgroup over both boxes (13 waves
# instead of 14) could save: half of the difference.   usage: tools/dev/detect_6waves.sh
cd "$(dirname "$0")/../.."
P=card.io-dmz_amd; D=gpurun_out/det6; mkdir -p $D
HF="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -I$P/csrc"
/opt/rocm/bin/hipcc $HF -DDMZ_DEV_HZ_LANES=372 -x hip -c $P/csrc/capi.cpp -o $D/capi.o 2>/dev/null
/opt/rocm/bin/hipcc $HF -DDMZ_DEV_HZ_LANES=372 -c $P/csrc/detect.hip -o $D/detect.o 2>/dev/null
OBJS=""
for f in geometry warp vseg hseg digits expiry session plumbing synth weights_blob; do OBJS="$OBJS $P/csrc/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/lib6.so $OBJS $D/capi.o $D/detect.o -ldl
for rep in 1 2 3; do
  for L in "" $PWD/$D/lib6.so; do
    echo -n "$( [ -z "$L" ] && echo 'seven waves (389 columns)' || echo 'six waves (372 columns)  ' ): "
    DMZ_HIP_LIB=$L python tools/stage_times.py 65536 3 2>/dev/null | grep -o "detect [0-9.]*"
  done
done
cd /tmp; export TMPDIR=/tmp
for L in "" $OLDPWD/$D/lib6.so; do
  rm -rf $OLDPWD/$D/ks; DMZ_HIP_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$D/ks -o ks -- python3 $OLDPWD/tools/stage_times.py 65536 3 > /dev/null 2>&1
  python3 - "$OLDPWD/$D/ks" "$L" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/ks_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_detect_walk" in r["Name"]:
            print("%s %s avg %.3f ms" % ("six waves " if sys.argv[2] else "seven waves", "hz  " if "false" in r["Name"] else "vert", float(r["AverageNs"]) / 1e6))
    break
PY
done
