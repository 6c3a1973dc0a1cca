#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_g; mkdir -p $O; cd $R
run() { env "$@" timeout 120 python tools/roofline_line.py 2>/dev/null | grep '^{' >> $O/lines.jsonl; tail -1 $O/lines.jsonl | cut -c1-330; }
: > $O/lines.jsonl
run MI_X=base
run MI_ROI_ALIGN_BWD_BATCH=3
run MI_X=base2
run MI_ROI_ALIGN_BWD_BATCH=3 MI_X=again
MI_ROI_ALIGN_BWD_BATCH=3 timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "roi_align" > $O/pytest_s3.log 2>&1; tail -3 $O/pytest_s3.log
