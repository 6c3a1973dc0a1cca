#!/bin/bash
# round-3 iteration aid: RoIAlign tests + the roofline object for the tile-centric forward and the record path
TAG=${1:-r3a}; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "roi_align or fpn or module" > gpurun_out/${TAG}_pytest.log 2>&1
tail -15 gpurun_out/${TAG}_pytest.log
for cfg in "X=1" "MI_ROI_ALIGN_IMPL=records" "X=1"; do
  echo "== $cfg" >> gpurun_out/${TAG}_roof.log
  env $cfg timeout 300 python bench.py --only-roofline 2>&1 | grep -v amdgpu.ids >> gpurun_out/${TAG}_roof.log
done
cat gpurun_out/${TAG}_roof.log
