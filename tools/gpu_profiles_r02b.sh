#!/bin/bash
# Round-2 profiles, second pass (after the fused AffineChannel pass, the fused selection kernels and the detection
# hipGraph): the judged bench line, the steady-state kernel table of the training step, its MFMA utilisation, the
# config-2 kernel durations and the kernel table of the inference glue.  The PMC passes over the RoIAlign kernels and the
# FETCH_SIZE calibration of tools/gpu_profiles_r02.sh are not repeated (profiles/r02_pmc_*, r02_fetch_size_calibration).
# Run on the GPU box through gpurun; output gpurun_out/prof_r02b/, copy what is judged into profiles/.
TAG=r02; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_${TAG}b; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 900 python bench.py 2> $O/bench_plain.err | grep '^{' | tail -1 > $O/bench_line.json)
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o t -f csv -- python $R/bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 > $O/trace_stdout.log 2>&1
cp $(find $O/trace -name '*kernel_stats.csv' | head -1) $O/train_step_kernel_stats_whole_process.csv
python $R/tools/trace_window.py $(find $O/trace -name '*kernel_trace.csv' | head -1) 5 > $O/train_step_steady_state.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d $O/pmc_mfma -o p -- python $R/bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 2 > $O/pmc_mfma_stdout.log 2>&1
python $R/tools/rocpd_mfma.py $(find $O/pmc_mfma -name '*.db') > $O/train_step_mfma_util.txt 2>&1
echo "pass,kernel,calls,avg_ns,min_ns,max_ns" > $O/config2_kernel_durations.csv
for pass in roi_align_fwd roi_align_bwd nhwc_fwd; do
  if [ $pass = nhwc_fwd ]; then export MI_BENCH_NHWC=1; k=roi_align_fwd; else unset MI_BENCH_NHWC; k=$pass; fi
  timeout 120 rocprofv3 --kernel-trace --stats -d $O/c2_$pass -o c -f csv -- python $R/tools/run_one_kernel.py $k 50 > $O/c2_$pass.log 2>&1
  python - $O/c2_$pass $pass >> $O/config2_kernel_durations.csv <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "roi_align" in row["Name"]:
            name = row["Name"].replace("void ", "").replace("mi::(anonymous namespace)::", "").split("(")[0]
            print("%s,\"%s\",%s,%.1f,%s,%s" % (sys.argv[2], name, row["Calls"], float(row["AverageNs"]), row["MinNs"], row["MaxNs"]))
PY
done
unset MI_BENCH_NHWC
GRAFT_GLUE_OUT=$O bash $R/tools/gpu_r2_glue.sh > $O/inference_glue_kernels.txt 2>&1
rm -rf $O/trace $O/pmc_mfma $O/c2_*/
cut -c1-700 $O/bench_line.json; echo; head -30 $O/train_step_steady_state.txt | cut -c1-150; tail -8 $O/train_step_steady_state.txt; head -12 $O/train_step_mfma_util.txt; cat $O/config2_kernel_durations.csv; cat $O/inference_glue_kernels.txt | cut -c1-150
