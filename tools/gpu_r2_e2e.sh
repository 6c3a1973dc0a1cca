#!/bin/bash
# Round-2 end-to-end measurements on the GPU box: the judged bench line, rocprofv3 kernel-trace stats of the eagerly
# launched training step, one PMC pass for the MFMA utilisation of the convolution kernels.  Output: gpurun_out/<TAG>/.
TAG=${1:-r02_e2e}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 900 python bench.py 2> $O/bench.err | grep '^{' | tail -1 > $O/bench_line.json); tail -3 $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o t -f csv -- python $R/bench.py --no-extras --no-cpu-baseline --launch eager --steps 10 --warmup 3 > $O/trace_stdout.log 2>&1
cp $O/trace/*kernel_stats.csv $O/train_step_kernel_stats.csv 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d $O/pmc -o p -- python $R/bench.py --no-extras --no-cpu-baseline --launch eager --steps 3 --warmup 2 > $O/pmc_stdout.log 2>&1
python $R/tools/rocpd_mfma.py $O/pmc/*.db > $O/train_step_mfma_util.txt 2>&1 || python $R/tools/rocpd_mfma.py $O/pmc/*/*.db > $O/train_step_mfma_util.txt 2>&1
rm -rf $O/trace $O/pmc
cut -c1-1500 $O/bench_line.json; echo; head -25 $O/train_step_kernel_stats.csv | cut -c1-200; head -30 $O/train_step_mfma_util.txt
