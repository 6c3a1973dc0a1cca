#!/bin/bash
# kernel durations of the library's kernels inside the real training step (rocprofv3 kernel trace, whole process: 13 steps)
TAG=${1:-r03}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/step_${TAG}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o t -f csv -- python $R/bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 > $O/stdout.log 2>&1
f=$(find $O/trace -name '*kernel_stats.csv' | head -1); cp $f $O/train_step_kernel_stats.csv
python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms over the process" % (tot / 1e6))
for r in rows:
    n = r["Name"]
    if "mi::" in n or "roi_align" in n:
        short = n.replace("void ", "").replace("mi::(anonymous namespace)::", "").split("(")[0]
        print("%-46s calls %5s avg %9.1f us  min %8.1f  max %8.1f  share %5.2f%%" % (short[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
grep '^{' $O/stdout.log | tail -1 | cut -c1-200
rm -rf $O/trace
