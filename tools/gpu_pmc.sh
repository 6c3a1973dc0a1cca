#!/bin/bash
# PMC passes over one kernel (separate rocprofv3 runs per counter group; never combined with sys/hip tracing)
# usage: bash tools/gpu_pmc.sh TAG KERNEL [ENV=VAL ...]
TAG=$1; K=$2; shift; shift
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/${TAG}_pmc$i -o p -- python $R/tools/run_one_kernel.py $K 5 > $R/gpurun_out/${TAG}_pmc$i.log 2>&1
  echo "group $i exit $?" >> $R/gpurun_out/${TAG}_pmc$i.log
done
python $R/tools/rocpd_pmc.py $R/gpurun_out/${TAG}_pmc*/*.db | grep -v "Fill\|distribution\|elementwise"
