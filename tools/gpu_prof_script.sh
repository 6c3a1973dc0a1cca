#!/bin/bash
# rocprofv3 kernel-trace statistics of the library's kernels for one python command; usage: bash tools/gpu_prof_script.sh TAG python tools/x.py ...
TAG=$1; shift; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_${TAG}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o t -f csv -- "$@" > $O/stdout.log 2>&1)
f=$(find $O/trace -name '*kernel_stats.csv' | head -1); cp $f $O/kernel_stats.csv
python - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "mi::" in n or "roi_align" in n:
        short = n.replace("void ", "").replace("mi::(anonymous namespace)::", "").split("(")[0]
        print("%-46s calls %5s avg %9.1f us  min %8.1f  max %8.1f" % (short[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
tail -2 $O/stdout.log | cut -c1-300
rm -rf $O/trace
