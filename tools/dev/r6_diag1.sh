#!/bin/bash
# GPU box: diagnostics of k_expiry_seg at HEAD: phase timeline and the cost of the library-order pick
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6_diag1
O=gpurun_out/r6_diag1
bash tools/dev/variant.sh expiry.hip -DDMZ_XSEG_TL tools/dev/xseg_tl.py > $O/xseg_tl.txt 2>&1
bash tools/dev/variant.sh expiry.hip -DDMZ_XSEG_DBG tools/dev/xseg_dbg.py > $O/xseg_dbg.txt 2>&1
