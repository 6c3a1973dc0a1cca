#!/bin/bash
# forward tuning iteration: RoIAlign parity tests, then bench --only-roofline for each "ENV=VAL ..." argument
# usage: bash tools/gpu_fwd_iter.sh TAG "A=1" "B=2 C=3" ...
TAG=${1:-i}; shift; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd $R
timeout 240 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "roi_align" --timeout 60 > $R/gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?" >> $R/gpurun_out/${TAG}_pytest.log
: > $R/gpurun_out/${TAG}_exp.log
for cfg in "$@"; do
  env $cfg timeout 300 python bench.py --only-roofline 2>&1 | grep -v amdgpu.ids >> $R/gpurun_out/${TAG}_exp.log
done
tail -5 $R/gpurun_out/${TAG}_pytest.log; cut -c1-420 $R/gpurun_out/${TAG}_exp.log
