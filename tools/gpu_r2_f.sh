#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_f; mkdir -p $O; cd $R
for p in proposals labelling fwd_only; do REPLAYS=10 timeout 150 python tools/capture_probe.py $p > $O/cap_$p.log 2>&1; echo "$p: $(grep -h 'CAPTURE_\|fault' $O/cap_$p.log | head -2 | tr '\n' ' ')"; done
# ablation of the backward tile kernel (tuning build: the MI_ROI_ALIGN_ABLATE switches are compiled in)
MI_TUNING_BUILD=1 python -m detectron_pytorch_amd.build --force > $O/build.log 2>&1; tail -1 $O/build.log
run() { env "$@" timeout 120 python tools/roofline_line.py 2>/dev/null | grep '^{' >> $O/lines.jsonl; tail -1 $O/lines.jsonl | cut -c1-200; }
: > $O/lines.jsonl
for a in 0 1 2 3; do run MI_ROI_ALIGN_BWD_BATCH=1 MI_ROI_ALIGN_ABLATE=$a; done
