#!/bin/bash
# GPU box: bench lines with the r6_v1 counters, the k_homography fault repro in its three builds, the fused expiry kernel's tests
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_diag4; mkdir -p $O
bash tools/bench_lines.sh r6_v1 > $O/bench_lines.txt 2>&1
for m in nocheck selfcheck check; do TAIL=30 bash tools/dev/homography_fault.sh $m ${REPS:-16} 8 >> $O/hfault.txt 2>&1; done
bash tools/dev/variant_lib.sh fused expiry.hip -DDMZ_XSEG_FUSED=1 > $O/fused_build.txt 2>&1
DMZ_HIP_LIB=$PWD/gpurun_out/ab/lib_fused.so timeout 900 python -m pytest tests/test_gpu_expiry.py tests/test_gpu_pipeline.py tests/test_gpu_parity_large.py tests/test_gpu_full_size.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 > $O/fused_tests.txt
