#!/bin/bash
# A/B of RoIAlign tuning variants inside ONE box (timings differ between boxes): one line per variant
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_perf; mkdir -p $O; cd $R
run() { env "$@" timeout 120 python tools/roofline_line.py 2>/dev/null | grep '^{' >> $O/lines.jsonl; tail -1 $O/lines.jsonl | cut -c1-400; }
: > $O/lines.jsonl
run MI_X=base
run MI_ROI_ALIGN_BWD_BATCH=16
run MI_ROI_ALIGN_FWD_GROUP=1
run MI_ROI_ALIGN_FWD_GROUP=2
run MI_ROI_ALIGN_NHWC_V=1
run MI_ROI_ALIGN_NHWC_V=1 MI_ROI_ALIGN_FWD_GROUP=2
run MI_ROI_ALIGN_BWD_TH=8
run MI_X=base2
