#!/bin/bash
# kernel-trace of the eagerly launched training step -> gpurun_out/trace_step/{kernel_stats.csv,steady_state.txt}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_step; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o t -f csv -- python $R/bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 $BENCH_ARGS > $O/stdout.log 2>&1
cp $(find $O/trace -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
python $R/tools/trace_window.py $(find $O/trace -name '*kernel_trace.csv' | head -1) 5 > $O/steady_state.txt 2>&1
rm -rf $O/trace
cat $O/steady_state.txt; grep '^{' $O/stdout.log | cut -c1-300
