#!/bin/bash
# kernel-trace stats of the RoIAlign forward launches only (tools/run_one_kernel.py); usage: bash tools/gpu_prof_fwd.sh TAG [ENV=VAL ...]
TAG=$1; shift; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; for kv in "$@"; do export "$kv"; done
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o p -f csv -- python $R/tools/run_one_kernel.py ${KERNEL:-roi_align_fwd} 50 > $R/gpurun_out/${TAG}_prof.log 2>&1
cat $R/gpurun_out/${TAG}_prof/*kernel_stats.csv 2>/dev/null | cut -c1-200 | head -8
