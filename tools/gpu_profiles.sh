#!/bin/bash
# The round's profiles (ROUND=r05 bash tools/gpu_profiles.sh ...): the judged bench line; rocprofv3 kernel statistics of the training step (steady-state window); the
# MFMA utilisation (PMC pass) of the training step AND of BASELINE config 5 (X-101-64x4d: grouped convolutions); the
# config-2 kernel durations of every RoIAlign forward / backward variant.  Run on the GPU box through gpurun; output
# gpurun_out/prof_$ROUND/, copy what is judged into profiles/.   usage: [ROUND=r05] bash tools/gpu_profiles.sh [all|bench|trace|mfma|config2|pmc|poolcrop]
ROUND=${ROUND:-r05}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$ROUND; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
STAGE=${1:-all}
if [ $STAGE = all ] || [ $STAGE = bench ]; then
(cd $R && timeout 1200 python bench.py 2> $O/bench_plain.err | grep '^{' | tail -1 > $O/bench_line.json)
fi
if [ $STAGE = all ] || [ $STAGE = trace ]; then
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o t -f csv -- python $R/bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 > $O/trace_stdout.log 2>&1
cp $(find $O/trace -name '*kernel_stats.csv' | head -1) $O/train_step_kernel_stats_whole_process.csv
python $R/tools/trace_window.py $(find $O/trace -name '*kernel_trace.csv' | head -1) 5 > $O/train_step_steady_state.txt 2>&1
rm -rf $O/trace
fi
if [ $STAGE = all ] || [ $STAGE = mfma ]; then
# counters in their own run (--kernel-trace only beside --pmc).  MIOpen's find step must not run under the profiler (its
# timings are distorted there and it settles on the naive_conv_* solvers: a first attempt showed 0.2 % MFMA and 70 % of the
# time in naive_conv): each workload runs once un-profiled first, which leaves the chosen solvers in MIOpen's user find-db.
(cd $R && timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 2 > $O/warm_step.log 2>&1)
(cd $R && timeout 900 python bench.py --only-config5 --steps 3 > $O/warm_config5.log 2>&1)
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d $O/pmc_mfma -o p -- python $R/bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 2 > $O/pmc_mfma_stdout.log 2>&1
python $R/tools/rocpd_mfma.py $(find $O/pmc_mfma -name '*.db') > $O/train_step_mfma_util.txt 2>&1
rm -rf $O/pmc_mfma
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d $O/pmc_mfma5 -o p -- python $R/bench.py --only-config5 --steps 3 > $O/pmc_mfma5_stdout.log 2>&1
python $R/tools/rocpd_mfma.py $(find $O/pmc_mfma5 -name '*.db') > $O/config5_mfma_util.txt 2>&1
rm -rf $O/pmc_mfma5
fi
if [ $STAGE = all ] || [ $STAGE = config2 ]; then
echo "pass,kernel,calls,avg_ns,min_ns,max_ns" > $O/config2_kernel_durations.csv
for pass in roi_align_fwd roi_align_fwd_records roi_align_bwd roi_align_bwd_unplanned nhwc_fwd; do
  unset MI_BENCH_NHWC MI_ROI_ALIGN_IMPL MI_BENCH_TILES_WS MI_BENCH_BWD_UNPLANNED MI_ROI_ALIGN_SLAB; k=roi_align_fwd
  case $pass in
    roi_align_fwd_records) export MI_ROI_ALIGN_SLAB=0;;   # the record-driven pair the records-free kernel replaced
    roi_align_bwd) k=roi_align_bwd;;
    roi_align_bwd_unplanned) k=roi_align_bwd; export MI_BENCH_BWD_UNPLANNED=1;;
    nhwc_fwd) export MI_BENCH_NHWC=1;;
  esac
  timeout 120 rocprofv3 --kernel-trace --stats -d $O/c2_$pass -o c -f csv -- python $R/tools/run_one_kernel.py $k 2000 > $O/c2_$pass.log 2>&1
  python - $O/c2_$pass $pass >> $O/config2_kernel_durations.csv <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "roi_align" in row["Name"]:
            name = row["Name"].replace("void ", "").replace("mi::(anonymous namespace)::", "").split("(")[0]
            print("%s,\"%s\",%s,%.1f,%s,%s" % (sys.argv[2], name, row["Calls"], float(row["AverageNs"]), row["MinNs"], row["MaxNs"]))
PY
  rm -rf $O/c2_$pass
done
unset MI_BENCH_NHWC MI_ROI_ALIGN_IMPL MI_BENCH_TILES_WS MI_BENCH_BWD_UNPLANNED MI_ROI_ALIGN_SLAB
fi
if [ $STAGE = all ] || [ $STAGE = pmc ]; then
for variant in records bwd; do
  k=roi_align_fwd; [ $variant = bwd ] && k=roi_align_bwd
  j=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" \
             "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
             "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE"; do
    j=$((j+1))
    timeout 120 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_${variant}_$j -o p -- python $R/tools/run_one_kernel.py $k 5 > $O/pmc_${variant}_$j.log 2>&1
  done
  python $R/tools/rocpd_pmc.py --json $O/pmc_$variant.json $O/pmc_${variant}_*/*.db | grep -v "Fill\|distribution\|elementwise" | cut -c1-120 > $O/pmc_$variant.txt
  rm -rf $O/pmc_${variant}_*/
done
unset MI_ROI_ALIGN_IMPL
fi
if [ $STAGE = all ] || [ $STAGE = poolcrop ]; then
# RoIPool / RoICrop backward tile kernels (round 6): durations alone, then the counter groups
echo "pass,kernel,calls,avg_ns,min_ns,max_ns" > $O/pool_crop_bwd_kernel_durations.csv
for k in roi_pool_bwd roi_crop_bwd; do
  timeout 120 rocprofv3 --kernel-trace --stats -d $O/pc_$k -o c -f csv -- python $R/tools/run_one_kernel.py $k 50 > $O/pc_$k.log 2>&1
  python - $O/pc_$k $k >> $O/pool_crop_bwd_kernel_durations.csv <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "roi_pool" in row["Name"] or "roi_crop" in row["Name"]:
            name = row["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            print("%s,\"%s\",%s,%.1f,%s,%s" % (sys.argv[2], name, row["Calls"], float(row["AverageNs"]), row["MinNs"], row["MaxNs"]))
PY
  rm -rf $O/pc_$k
  j=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" \
             "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
             "SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
    j=$((j+1))
    timeout 120 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_${k}_$j -o p -- python $R/tools/run_one_kernel.py $k 5 > $O/pmc_${k}_$j.log 2>&1
  done
  python $R/tools/rocpd_pmc.py --json $O/pmc_$k.json $O/pmc_${k}_*/*.db | grep "roi_pool\|roi_crop" | cut -c1-120 > $O/pmc_$k.txt
  rm -rf $O/pmc_${k}_*/ $O/pmc_${k}_*.log
done
(cd $R && python tools/pool_crop_time.py 100 > $O/pool_crop_time.json 2>/dev/null)
fi
cut -c1-600 $O/bench_line.json 2>/dev/null; echo; head -24 $O/train_step_steady_state.txt 2>/dev/null | cut -c1-150; head -14 $O/train_step_mfma_util.txt $O/config5_mfma_util.txt 2>/dev/null | cut -c1-130; cat $O/config2_kernel_durations.csv 2>/dev/null; cat $O/pmc_records.txt $O/pmc_bwd.txt $O/pool_crop_bwd_kernel_durations.csv $O/pmc_roi_pool_bwd.txt $O/pmc_roi_crop_bwd.txt 2>/dev/null
