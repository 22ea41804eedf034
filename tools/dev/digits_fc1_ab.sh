#!/bin/bash
# GPU box: A/B of k_digits variants that need capi.cpp (the weight fragments it uploads) AND digits.hip rebuilt together, e.g. the
# first fully connected layer on f32 matrix instructions (-DDMZ_DG_FC1_F16=0: 24 x 16x16x4 f32 per wave and column) against the
# split-f16 form (9 x 16x16x32 f16).  usage: tools/dev/digits_fc1_ab.sh <outdir-name> "<flags of variant 0>" "<flags of variant 1>" ...
# DG_TEST="<pytest files>" runs them against the LAST variant first.
cd "$(dirname "$0")/../.."
P=card.io-dmz_amd; O=gpurun_out/$1; shift; mkdir -p $O
HF="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -I$P/csrc"
N=0
for FL in "$@"; do
  /opt/rocm/bin/hipcc $HF $FL -x hip -c $P/csrc/capi.cpp -o $O/capi$N.o 2>/dev/null
  /opt/rocm/bin/hipcc $HF $FL -c $P/csrc/digits.hip -o $O/digits$N.o 2>/dev/null
  OBJS=""
  for f in detect geometry warp vseg hseg expiry session plumbing synth weights_blob; do OBJS="$OBJS $P/csrc/$f.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/lib_v$N.so $OBJS $O/capi$N.o $O/digits$N.o -ldl
  echo "variant $N: $FL"
  N=$((N+1))
done > $O/variants.txt
N=$((N-1))
if [ -n "$DG_TEST" ]; then DMZ_HIP_LIB=$PWD/$O/lib_v$N.so timeout 1500 python -m pytest $DG_TEST -x -q -m gpu 2>&1 | grep -v "NCCL\|RCCL" | tail -6 > $O/test.txt; fi
for rep in 1 2 3; do for V in $(seq 0 $N); do
  echo -n "variant $V: "; DMZ_HIP_LIB=$PWD/$O/lib_v$V.so python tools/stage_times.py 65536 2 2>/dev/null | cut -d" " -f3-
done; done > $O/stages.txt 2>&1
for rep in 1 2 3; do for V in $(seq 0 $N); do
  echo -n "variant $V: "
  DMZ_HIP_LIB=$PWD/$O/lib_v$V.so python bench.py --no-cpu-baseline 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms per step', d['value'], 'frames/s')"
done; done > $O/step.txt 2>&1
cat $O/variants.txt $O/test.txt $O/stages.txt $O/step.txt
