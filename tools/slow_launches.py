#!/usr/bin/env python3
"""From a rocprofv3 --hip-trace --kernel-trace CSV pair: the HIP API calls that took longer than MS (default 20) and the
kernel each launched (joined by correlation id).  usage: python tools/slow_launches.py DIR [MS]"""
import csv
import glob
import sys

d, ms = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
after_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0  # ignore calls that start earlier than this (warm-up)
api = glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0]
ker = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
names = {}
for r in csv.DictReader(open(ker)):
    names[r["Correlation_Id"]] = r["Kernel_Name"]
rows = sorted(csv.DictReader(open(api)), key=lambda r: int(r["Start_Timestamp"]))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
prev = None
for r in rows:
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if dur >= ms and (int(r["Start_Timestamp"]) - t0) / 1e6 >= after_ms:
        print("%9.1f ms  at %9.1f ms  tid %s  %-28s %s   (previous call: %s)" % (
            dur, (int(r["Start_Timestamp"]) - t0) / 1e6, r.get("Thread_Id", "?"), r["Function"],
            names.get(r["Correlation_Id"], "?")[:110], prev))
    prev = "%s %s (tid %s)" % (r["Function"], names.get(r["Correlation_Id"], "")[:60], r.get("Thread_Id", "?"))

# per interval between two hipDeviceSynchronize calls (one test image in tools/stall_probe.py): wall time, time inside HIP
# API calls, number of calls, the three longest calls
print("\nintervals between hipDeviceSynchronize calls:")
marks = [i for i, r in enumerate(rows) if r["Function"] == "hipDeviceSynchronize"]
for a, b in zip(marks[:-1], marks[1:]):
    seg = rows[a + 1:b + 1]
    if len(seg) < 50:
        continue
    wall = (int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e6
    durs = sorted(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["Function"],
                   names.get(r["Correlation_Id"], "")[:40]) for r in seg)
    inside = sum(d[0] for d in durs)
    gaps = sorted((((int(y["Start_Timestamp"]) - int(x["End_Timestamp"])) / 1e6, x["Function"], y["Function"])
                   for x, y in zip(seg[:-1], seg[1:]) if x.get("Thread_Id") == y.get("Thread_Id")), reverse=True)[:2]
    print("wall %7.1f ms  in HIP calls %7.1f ms  calls %5d  longest: %s | largest gaps between calls: %s" % (
        wall, inside, len(seg), "; ".join("%.1f %s %s" % d for d in durs[-3:]), "; ".join("%.1f ms after %s before %s" % g for g in gaps)))
