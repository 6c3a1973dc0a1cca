#!/bin/bash
# Round profiles: rocprofv3 kernel-trace stats of `python bench.py` (the judged command) + PMC passes (separate runs)
# over the roofline kernel.  Writes summaries under gpurun_out/prof_<TAG>/ ; copy what is to be judged into profiles/.
TAG=${1:-r01}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/bench -o b -f csv -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_stdout.log 2>&1
cp $O/bench/*kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
j=0
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_EA0_RDREQ_sum" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "TA_BUSY_avr TA_BUSY_max TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
  j=$((j+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc$j -o p -- python $R/tools/run_one_kernel.py roi_align_fwd 5 > $O/pmc$j.log 2>&1
  timeout 120 rocprofv3 --kernel-trace --pmc $grp -d $O/pmcb$j -o p -- python $R/tools/run_one_kernel.py roi_align_bwd 5 > $O/pmcb$j.log 2>&1
done
python $R/tools/rocpd_pmc.py $O/pmc*/*.db | grep -v "Fill\|distribution\|elementwise" > $O/pmc_summary.txt
tail -3 $O/bench_stdout.log | cut -c1-300; head -12 $O/bench_kernel_stats.csv | cut -c1-160; grep -c . $O/pmc_summary.txt
