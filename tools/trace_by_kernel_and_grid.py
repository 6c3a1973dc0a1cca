#!/usr/bin/env python3
"""Average duration of every roi_* kernel of a rocprofv3 --kernel-trace run, keyed by kernel name AND grid size (one name
serves several shapes in a run of tools/bwd_clustered.py).  usage: python tools/trace_by_kernel_and_grid.py <dir of the trace>"""
import collections
import csv
import glob
import sys

for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "roi_" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void mi::", "").replace("(anonymous namespace)::", "")[:52]
        grid = r.get("Grid_Size_X") or r.get("Grid_Size") or ""
        d[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(d.items()):
        print("%-54s grid %-9s calls %4d  avg %8.1f us" % (k[0], k[1], len(v), sum(v) / len(v)))
