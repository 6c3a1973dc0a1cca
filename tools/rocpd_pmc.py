#!/usr/bin/env python3
"""Print per-kernel PMC counter averages from rocprofv3 rocpd databases: python tools/rocpd_pmc.py db [db ...]"""
import sqlite3
import sys

for path in sys.argv[1:]:
    c = sqlite3.connect(path)
    try:
        cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    except Exception as e:
        print(path, "no counters", e)
        continue
    namecol = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else None)
    q = ("select %s, counter_name, count(*), avg(value) from counters_collection group by 1, 2 order by 1, 2" % namecol)
    for kname, cname, n, avg in c.execute(q):
        if "at6native" in (kname or "") or "rocclr" in (kname or ""):
            continue
        print("%-50s %-26s n=%-4d avg=%.4g" % ((kname or "")[-50:], cname, n, avg))
