#!/bin/bash
# round-3 iteration aid: shape checks against the oracle, phase timeline, ablation masks (tuning build) of the tile forward
TAG=${1:-r3}; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd $R
timeout 300 python tools/fwd_tiles_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_check.log; grep -v "^\.\." gpurun_out/${TAG}_check.log | tail -14
timeout 120 python tools/timeline_tiles.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_tl.log; cat gpurun_out/${TAG}_tl.log
for m in ${MASKS:-0 1 2 4 7 0}; do
  MI_ROI_ALIGN_ABLATE=$m timeout 120 python tools/fwd_time.py 2>&1 | grep -v amdgpu.ids | sed "s/^/ablate=$m /" >> gpurun_out/${TAG}_abl.log
done
cat gpurun_out/${TAG}_abl.log
