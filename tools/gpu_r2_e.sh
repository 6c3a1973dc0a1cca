#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_e; mkdir -p $O; cd $R
run() { env "$@" timeout 120 python tools/roofline_line.py 2>/dev/null | grep '^{' >> $O/lines.jsonl; tail -1 $O/lines.jsonl | cut -c1-330; }
: > $O/lines.jsonl
run MI_X=v2
run MI_ROI_ALIGN_BWD_BATCH=1
run MI_X=v2b
timeout 400 python -m pytest tests/test_ops_gpu.py tests/test_e2e_gpu.py -m gpu -q > $O/pytest.log 2>&1; tail -6 $O/pytest.log
REPLAYS=10 timeout 200 python tools/capture_probe.py body_fwd_bwd > $O/cap_body.log 2>&1; tail -3 $O/cap_body.log
REPLAYS=10 timeout 200 python tools/capture_probe.py fwd_bwd > $O/cap_fwd_bwd.log 2>&1; tail -4 $O/cap_fwd_bwd.log
