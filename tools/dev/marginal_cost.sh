#!/bin/bash
# Developer tool (GPU box): what each kernel costs INSIDE the three-queue pipeline.  For every kernel tag a library that
# launches that kernel twice (-DDMZ_DUP=<tag>, dmz_hip_internal.h); the step time minus the plain library's is the kernel's
# marginal cost beside whatever runs on the other queues.  usage: tools/dev/marginal_cost.sh build (here), then on the GPU box tools/dev/marginal_cost.sh [batch] [reps]
cd "$(dirname "$0")/../.."
declare -A FILE=( [detect_h]=detect.hip [detect_v]=detect.hip [warp]=warp.hip [vseg]=vseg.hip [hseg]=hseg.hip [patches]=digits.hip
                  [digits]=digits.hip [stripes]=expiry.hip [xseg]=expiry.hip [xcat]=expiry.hip )
TAGS="detect_h detect_v warp vseg hseg patches digits stripes xseg xcat"
export DMZ_AB_DIR=ab_libs
if [ "$1" == "build" ]; then  # (in the build container: the libraries travel with the snapshot)
  for t in $TAGS; do bash tools/dev/variant_lib.sh dup_$t ${FILE[$t]} -DDMZ_DUP=$t || exit 1; done
  exit 0
fi
B=${1:-65536}; R=${2:-8}
for round in 1 2; do
  echo "base      $(python tools/dev/pipe_ab.py $B $R)"
  for t in $TAGS; do
    echo "$(printf '%-9s' $t) $(DMZ_HIP_LIB=$PWD/ab_libs/lib_dup_$t.so python tools/dev/pipe_ab.py $B $R)"
  done
done
echo "base      $(python tools/dev/pipe_ab.py $B $R)"
