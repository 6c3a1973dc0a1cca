#!/bin/bash
# kernel durations (rocprofv3 --kernel-trace --stats) of the RoIAlign backward on the step's RoIs and config 2, planned
# launch; usage: bash tools/gpu_bwd_kernels.sh [tag]
TAG=${1:-r04}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bwd_${TAG}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
SLICES="${SLICES:-32}" timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o t -f csv -- python $R/tools/bwd_time.py 30 > $O/stdout.log 2>&1
f=$(find $O/trace -name '*kernel_stats.csv' | head -1); cp $f $O/bwd_kernel_stats.csv
python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if "roi_align" in n:
        short = n.replace("void ", "").replace("mi::(anonymous namespace)::", "").split("(")[0]
        print("%-52s calls %5s avg %9.1f us  min %8.1f  max %8.1f" % (short[:52], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
grep -v amdgpu.ids $O/stdout.log | tail -3
rm -rf $O/trace
