#!/bin/bash
# backward iteration: RoIAlign parity tests (TESTS=0 skips them), then tools/bwd_time.py for the slice lengths in SLICES
mkdir -p gpurun_out
if [ "${TESTS:-1}" = "1" ]; then
  timeout 1200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "roi_align or roi_feature or reference_modules" 2>&1 | tail -8
fi
SLICES="${SLICES:-0 16 32 64}" timeout 900 python tools/bwd_time.py 50 2>&1 | tail -8
