#!/bin/bash
# round-3 iteration aid: shape checks against the oracle, the phase timeline and the roofline object of the tile forward
TAG=${1:-r3}; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd $R
timeout 300 python tools/fwd_tiles_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_check.log; tail -14 gpurun_out/${TAG}_check.log
timeout 120 python tools/timeline_tiles.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_tl.log; cat gpurun_out/${TAG}_tl.log
timeout 300 python bench.py --only-roofline 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_roof.log; cat gpurun_out/${TAG}_roof.log
