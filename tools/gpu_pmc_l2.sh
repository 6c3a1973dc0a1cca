#!/bin/bash
# L2 / fabric counters of the RoIAlign forward kernel for a list of "ENV=VAL ..." configurations
TAG=$1; shift; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i+1)); j=0
  for grp in "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE"; do
    j=$((j+1))
    env $cfg timeout 120 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/${TAG}_c${i}_$j -o p -- python $R/tools/run_one_kernel.py roi_align_fwd 5 > $R/gpurun_out/${TAG}_c${i}_$j.log 2>&1
  done
  echo "== $cfg"; python $R/tools/rocpd_pmc.py $R/gpurun_out/${TAG}_c${i}_*/*.db | grep -v "Fill\|distribution\|elementwise\|prepare" | cut -c40-200
done
