#!/bin/bash
# A/B the RoIAlign forward variants on the config-2 shape (tuning aid)
TAG=${1:-s}; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "roi_align" > $R/gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?" >> $R/gpurun_out/${TAG}_pytest.log
: > $R/gpurun_out/${TAG}_sweep.log
for ring in 192 256 320; do
  MI_ROI_ALIGN_RING=$ring timeout 300 python bench.py --only-roofline >> $R/gpurun_out/${TAG}_sweep.log 2>&1
done
MI_ROI_ALIGN_IMPL=direct timeout 300 python bench.py --only-roofline >> $R/gpurun_out/${TAG}_sweep.log 2>&1
tail -4 $R/gpurun_out/${TAG}_pytest.log; cat $R/gpurun_out/${TAG}_sweep.log
