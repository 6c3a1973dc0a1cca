#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/glue; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for part in proposals detect; do
timeout 200 rocprofv3 --kernel-trace --stats -d $O/t_$part -o t -f csv -- python $R/tools/glue_probe.py 20 $part > $O/$part.log 2>&1
python - $(find $O/t_$part -name '*kernel_stats.csv' | head -1) $part <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("== %s: %.1f us of kernel time per image, %d launches per image" % (sys.argv[2], tot / 20 / 1e3, sum(int(r["Calls"]) for r in rows) / 20))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print("%7.1f us/img %5.1f calls/img %8.1f us avg  %s" % (float(r["TotalDurationNs"]) / 20 / 1e3, int(r["Calls"]) / 20, float(r["AverageNs"]) / 1e3, r["Name"][:130]))
PY
rm -rf $O/t_$part
done
