#!/bin/bash
# GPU box: k_expiry_cat's conv2 with the fences of its strict load -> wait -> matrix order removed (-DDMZ_C2_SCHED=1) against the
# shipped form: stage times, then run-to-run determinism of every record over repeated passes.   usage: tools/dev/c2sched_ab.sh [reps]
cd "$(dirname "$0")/../.."
REPS=${1:-8}
L1=$(bash tools/dev/variant_lib.sh c2s1 expiry.hip -DDMZ_C2_SCHED=1)
L2=$(bash tools/dev/variant_lib.sh c2kb2 expiry.hip -DDMZ_C2_SCHED=1 -DDMZ_C2_KB=2)
for rep in 1 2 3; do for V in "" $PWD/$L1 $PWD/$L2; do
  echo -n "${V:-shipped}: "; DMZ_HIP_LIB=$V python tools/stage_times.py 65536 2 2>/dev/null | grep -o "expiry_cat [0-9.]*"
done; done
for V in $PWD/$L1 $PWD/$L2; do echo "== determinism, $V"; DMZ_HIP_LIB=$V python tools/dev/determinism.py 65536 $REPS 2>&1 | tail -6; done
