#!/bin/bash
# Round-3 profiles: the judged bench line, the rocprofv3 kernel statistics of the same command (whole process and the
# steady-state window), the config-2 kernel durations of every RoIAlign forward / backward variant, and the PMC passes
# (memory traffic, L2 <-> L1 requests, unit utilisation) over the record forward and the tile-centric forward.
# Run on the GPU box through gpurun; output gpurun_out/prof_r03/, copy what is judged into profiles/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r03; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
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
if [ $STAGE = all ] || [ $STAGE = config2 ]; then
echo "pass,kernel,calls,avg_ns,min_ns,max_ns" > $O/config2_kernel_durations.csv
for pass in roi_align_fwd roi_align_bwd roi_align_bwd_unplanned nhwc_fwd tiles_one_launch tiles_descriptors; do
  unset MI_BENCH_NHWC MI_ROI_ALIGN_IMPL MI_BENCH_TILES_WS MI_BENCH_BWD_UNPLANNED; k=roi_align_fwd
  case $pass in
    roi_align_bwd) k=roi_align_bwd;;
    roi_align_bwd_unplanned) k=roi_align_bwd; export MI_BENCH_BWD_UNPLANNED=1;;
    nhwc_fwd) export MI_BENCH_NHWC=1;;
    tiles_one_launch) export MI_ROI_ALIGN_IMPL=tiles;;
    tiles_descriptors) export MI_ROI_ALIGN_IMPL=tiles MI_BENCH_TILES_WS=1;;
  esac
  timeout 120 rocprofv3 --kernel-trace --stats -d $O/c2_$pass -o c -f csv -- python $R/tools/run_one_kernel.py $k 50 > $O/c2_$pass.log 2>&1
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
unset MI_BENCH_NHWC MI_ROI_ALIGN_IMPL MI_BENCH_TILES_WS MI_BENCH_BWD_UNPLANNED
fi
if [ $STAGE = all ] || [ $STAGE = steprois ]; then
(cd $R && for c in box mask cfg2; do for sl in 0 32; do
  echo "=== case $c MI_ROI_ALIGN_BWD_SLICE=$sl"
  CASES=$c SLICES=$sl bash tools/gpu_prof_script.sh bwd_${c}_$sl python tools/bwd_time.py 30 | grep -v "simple_timer"
done; done) > $O/step_rois_backward_kernels.txt 2>&1
fi
if [ $STAGE = all ] || [ $STAGE = pmc ]; then
for variant in records tiles bwd; do
  unset MI_ROI_ALIGN_IMPL; [ $variant = tiles ] && export MI_ROI_ALIGN_IMPL=tiles
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
cut -c1-600 $O/bench_line.json 2>/dev/null; echo; head -24 $O/train_step_steady_state.txt 2>/dev/null | cut -c1-150; cat $O/config2_kernel_durations.csv $O/step_rois_backward_kernels.txt 2>/dev/null; cat $O/pmc_records.txt $O/pmc_tiles.txt $O/pmc_bwd.txt 2>/dev/null
