#!/usr/bin/env python3
"""Per-kernel PMC counter averages from rocprofv3 rocpd databases.
    python tools/rocpd_pmc.py db [db ...]                 # text lines: kernel counter n avg
    python tools/rocpd_pmc.py --json out.json db [...]    # {kernel: {counter: avg}} as well"""
import json
import re
import sqlite3
import sys


def short(name):
    """`void mi::(anonymous namespace)::roi_align_fwd_records<2, 336, 32, 1>(mi::LevelTable, ...)` ->
    `roi_align_fwd_records<2,336,32,1>`"""
    name = (name or "").replace("(anonymous namespace)::", "").replace("mi::", "")
    if name.startswith("void "):
        name = name[5:]
    m = re.match(r"([A-Za-z_][A-Za-z0-9_]*)(<[^()]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")).replace(" ", "") if m else name[-50:]


args = sys.argv[1:]
out_json = None
if args and args[0] == "--json":
    out_json, args = args[1], args[2:]
table = {}
for path in args:
    c = sqlite3.connect(path)
    try:
        cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    except Exception as e:
        print(path, "no counters", e)
        continue
    namecol = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else None)
    q = ("select %s, counter_name, count(*), avg(value) from counters_collection group by 1, 2 order by 1, 2" % namecol)
    for kname, cname, n, avg in c.execute(q):
        if "at6native" in (kname or "") or "rocclr" in (kname or "") or "at::native" in (kname or ""):
            continue
        s = short(kname)
        print("%-46s %-28s n=%-4d avg=%.4g" % (s[:46], cname, n, avg))
        table.setdefault(s, {})[cname] = avg
if out_json:
    with open(out_json, "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
