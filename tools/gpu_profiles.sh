#!/bin/bash
# Round profiles: rocprofv3 kernel-trace stats of `python bench.py` (the judged command) + PMC passes (separate runs,
# --kernel-trace only beside --pmc) over the roofline kernels.  Writes gpurun_out/prof_<TAG>/ ; copy what is to be
# judged into profiles/ (tools/make_pmc_json.py composes the JSON bench.py reads `traffic` from).
TAG=${1:-r01}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/bench -o b -f csv -- python $R/bench.py > $O/bench_stdout.log 2>&1
cp $O/bench/*kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
# the judged line comes from a plain run: the tracer's per-launch overhead inflates the launch-bound entries
(cd $R && timeout 900 python bench.py 2> $O/bench_plain.err | grep '^{' | tail -1 > $O/bench_line.json)
j=0
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_EA0_RDREQ_sum" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "TA_BUSY_avr TA_BUSY_max TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
  j=$((j+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc$j -o p -- python $R/tools/run_one_kernel.py roi_align_fwd 5 > $O/pmc$j.log 2>&1
  timeout 120 rocprofv3 --kernel-trace --pmc $grp -d $O/pmcb$j -o p -- python $R/tools/run_one_kernel.py roi_align_bwd 5 > $O/pmcb$j.log 2>&1
  MI_BENCH_NHWC=1 timeout 120 rocprofv3 --kernel-trace --pmc $grp -d $O/pmcn$j -o p -- python $R/tools/run_one_kernel.py roi_align_fwd 5 > $O/pmcn$j.log 2>&1
done
python $R/tools/rocpd_pmc.py --json $O/pmc_fwd.json $O/pmc[0-9]*/*.db > $O/pmc_fwd.txt
python $R/tools/rocpd_pmc.py --json $O/pmc_bwd.json $O/pmcb[0-9]*/*.db > $O/pmc_bwd.txt
python $R/tools/rocpd_pmc.py --json $O/pmc_nhwc.json $O/pmcn[0-9]*/*.db > $O/pmc_nhwc.txt
python $R/tools/make_pmc_json.py $TAG $O/pmc_fwd.json $O/pmc_bwd.json $O/pmc_nhwc.json > $O/pmc_roi_align.json
rm -rf $O/pmc[0-9]* $O/pmcb[0-9]* $O/pmcn[0-9]* $O/bench
cut -c1-400 $O/bench_line.json; head -14 $O/bench_kernel_stats.csv | cut -c1-160; grep hbm_bytes $O/pmc_roi_align.json
