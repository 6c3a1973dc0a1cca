#!/bin/bash
# scratch: PMC passes over the config-2 forward (records-free kernel by default)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_r06s; mkdir -p $O
j=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum" "TD_TD_BUSY_avr TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  j=$((j+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_slab_$j -o p -- python $R/tools/run_one_kernel.py roi_align_fwd 5 > $O/pmc_slab_$j.log 2>&1
done
python $R/tools/rocpd_pmc.py --json $O/pmc_slab.json $O/pmc_slab_*/*.db | grep -v "Fill\|distribution\|elementwise" | cut -c1-120 > $O/pmc_slab.txt
rm -rf $O/pmc_slab_*/
timeout 120 rocprofv3 --kernel-trace --stats -d $O/c2 -o c -f csv -- python $R/tools/run_one_kernel.py roi_align_fwd 50 > $O/c2.log 2>&1
grep -h "roi_align" $(find $O/c2 -name "*kernel_stats.csv") | cut -c1-200 > $O/slab_kernel_durations.csv; rm -rf $O/c2
cat $O/pmc_slab.txt $O/slab_kernel_durations.csv
