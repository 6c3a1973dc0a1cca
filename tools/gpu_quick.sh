#!/bin/bash
# quick GPU pass: parity tests + bench (+ optional rocprof with TAG); usage: bash tools/gpu_quick.sh TAG [prof]
TAG=${1:-q}; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd $R
timeout 1200 python -m pytest tests -m gpu -q -x > $R/gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?" >> $R/gpurun_out/${TAG}_pytest.log
timeout 300 python __graft_entry__.py --smoke > $R/gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke exit $?" >> $R/gpurun_out/${TAG}_smoke.log
timeout 600 python bench.py > $R/gpurun_out/${TAG}_bench.log 2>&1; echo "bench exit $?" >> $R/gpurun_out/${TAG}_bench.log
if [ "$2" = "prof" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o ${TAG} -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${TAG}_rocprof.log 2>&1
  echo "rocprof exit $?" >> $R/gpurun_out/${TAG}_rocprof.log
fi
tail -6 $R/gpurun_out/${TAG}_pytest.log; tail -2 $R/gpurun_out/${TAG}_smoke.log; tail -2 $R/gpurun_out/${TAG}_bench.log
