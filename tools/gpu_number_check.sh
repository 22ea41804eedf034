#!/bin/bash
# Developer tool (GPU box): number-path parity tests + per-stage timings
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_stages.py tests/test_golden_pipeline.py tests/test_gpu_parity_large.py -x -q -m gpu 2>&1 | tail -5
python tools/stage_times.py 8192 3
