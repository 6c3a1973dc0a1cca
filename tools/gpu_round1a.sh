#!/bin/bash
# first GPU pass: parity tests, smoke, bench, rocprofv3 kernel trace
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -8 > $R/gpurun_out/a_rocminfo.txt 2>&1
nproc >> $R/gpurun_out/a_rocminfo.txt; grep -m1 "model name" /proc/cpuinfo >> $R/gpurun_out/a_rocminfo.txt
cd $R
timeout 900 python -m pytest tests -m gpu -q > $R/gpurun_out/a_pytest.log 2>&1; echo "pytest exit $?" >> $R/gpurun_out/a_pytest.log
timeout 300 python __graft_entry__.py --smoke > $R/gpurun_out/a_smoke.log 2>&1; echo "smoke exit $?" >> $R/gpurun_out/a_smoke.log
timeout 600 python bench.py > $R/gpurun_out/a_bench.log 2>&1; echo "bench exit $?" >> $R/gpurun_out/a_bench.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/a_prof -o a -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/a_rocprof.log 2>&1
echo "rocprof exit $?" >> $R/gpurun_out/a_rocprof.log
tail -5 $R/gpurun_out/a_pytest.log; tail -3 $R/gpurun_out/a_smoke.log; tail -2 $R/gpurun_out/a_bench.log
