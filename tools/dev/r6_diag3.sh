#!/bin/bash
# GPU box: phase timelines of k_digits (-DDMZ_DG_TIMING) and k_expiry_cat (-DDMZ_XC_TIMING)
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_diag3; mkdir -p $O
bash tools/dev/variant.sh digits.hip -DDMZ_DG_TIMING tools/dev/dg_timing.py > $O/dg.txt 2>&1
bash tools/dev/variant.sh digits.hip "-DDMZ_DG_TIMING" "tools/dev/dg_timing.py 65536" >> $O/dg.txt 2>&1
bash tools/dev/variant.sh expiry.hip -DDMZ_XC_TIMING tools/dev/expiry_timing.py > $O/xc.txt 2>&1
