#!/bin/bash
# tuning aid: run a list of "ENV=VAL ... " configurations of bench.py --only-roofline; usage: bash tools/gpu_exp.sh TAG "A=1 B=2" "C=3" ...
TAG=$1; shift; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd $R
: > $R/gpurun_out/${TAG}_exp.log
for cfg in "$@"; do
  env $cfg timeout 300 python bench.py --only-roofline 2>&1 | grep -v amdgpu.ids >> $R/gpurun_out/${TAG}_exp.log
done
cat $R/gpurun_out/${TAG}_exp.log
