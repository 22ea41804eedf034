#!/bin/bash
# GPU box: tests + A/B of the working copy of expiry.hip against HEAD + the tie counters
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_diag2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_expiry.py tests/test_gpu_pipeline.py tests/test_gpu_parity_large.py tests/test_gpu_full_size.py -x -q -m gpu 2>&1 | tail -4 > $O/test.txt
AB_STAGES=1 bash tools/ab_bench.sh expiry.hip 3 > $O/stages.txt 2>&1
bash tools/ab_bench.sh expiry.hip 3 > $O/step.txt 2>&1
bash tools/dev/variant.sh expiry.hip -DDMZ_XSEG_DBG tools/dev/xseg_dbg.py > $O/xseg_dbg.txt 2>&1
