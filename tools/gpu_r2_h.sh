#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_h; mkdir -p $O; cd $R
run() { env "$@" timeout 120 python tools/roofline_line.py 2>/dev/null | grep '^{' >> $O/lines.jsonl; tail -1 $O/lines.jsonl | cut -c1-330; }
: > $O/lines.jsonl
run MI_X=a
run MI_X=b
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
