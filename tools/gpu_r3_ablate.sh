#!/bin/bash
# tuning build only (MI_TUNING_BUILD=1 python -m detectron_pytorch_amd.build --force): config-2 forward time per ablation mask
TAG=${1:-r3}; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd $R
for m in 0 1 2 4 3 6 7 0; do
  MI_ROI_ALIGN_ABLATE=$m timeout 120 python tools/fwd_time.py 2>&1 | grep -v amdgpu.ids | sed "s/^/ablate=$m /" >> gpurun_out/${TAG}_abl.log
done
cat gpurun_out/${TAG}_abl.log
