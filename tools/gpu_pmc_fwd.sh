#!/bin/bash
# utilisation counters of the RoIAlign forward kernels; usage: bash tools/gpu_pmc_fwd.sh TAG [ENV=VAL ...]
TAG=$1; shift; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
j=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE" \
           "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TD_BUSY_avr TCP_TA_TCP_STATE_READ_sum"; do
  j=$((j+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/${TAG}_u$j -o p -- python $R/tools/run_one_kernel.py roi_align_fwd 5 > $R/gpurun_out/${TAG}_u$j.log 2>&1
done
python $R/tools/rocpd_pmc.py $R/gpurun_out/${TAG}_u*/*.db | grep -v "Fill\|distribution\|elementwise" | cut -c38-200
