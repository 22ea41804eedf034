#!/bin/bash
# GPU box: k_digit_patches phase by phase (-DDMZ_DP_STOP=k returns after phase k: 1 strip staged, 2 gradients, 3 histograms, 4 LUTs;
# 99 = the whole kernel): busy cycles and LDS counters per dispatch, batch 8192.   usage: tools/dev/dp_phases.sh
cd "$(dirname "$0")/../.."
for S in 1 2 3 4 99; do
  L=$(bash tools/dev/variant_lib.sh dp$S digits.hip -DDMZ_DP_STOP=$S)
  echo "== DMZ_DP_STOP=$S"
  DMZ_HIP_LIB=$PWD/$L BATCH=8192 bash tools/pmc_kernels.sh SQ_BUSY_CU_CYCLES+SQ_LDS_IDX_ACTIVE+SQ_LDS_BANK_CONFLICT+SQ_ACTIVE_INST_VALU+SQ_INSTS_LDS+SQ_INSTS_VALU 2>&1 | grep "per dispatch\|k_digit_patches"
done
