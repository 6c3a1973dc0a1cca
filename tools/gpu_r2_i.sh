#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_i; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats -d $O/c2 -o c -f csv -- python $R/tools/run_one_kernel.py roi_align_bwd 50 > $O/c2.log 2>&1
python - $O/c2 <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "roi_align" in row["Name"]:
            print(row["Name"].replace("(anonymous namespace)::", "").split("(")[0][-40:], row["Calls"], row["AverageNs"])
PY
cd $R
run() { env "$@" timeout 120 python tools/roofline_line.py 2>/dev/null | grep '^{' >> $O/lines.jsonl; tail -1 $O/lines.jsonl | cut -c1-330; }
: > $O/lines.jsonl
run MI_X=a
rm -rf $O/c2
