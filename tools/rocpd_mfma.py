#!/usr/bin/env python3
"""MFMA utilisation per kernel from a rocprofv3 PMC pass (rocpd sqlite) that collected SQ_VALU_MFMA_BUSY_CYCLES,
SQ_BUSY_CU_CYCLES and GRBM_GUI_ACTIVE beside --kernel-trace.

    python tools/rocpd_mfma.py <results.db> [...]

util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 256 CUs * 4 SIMDs): the fraction of SIMD-cycles of the dispatch's
wall time in which a matrix pipe was busy (the gfx94x MfmaUtil formula; ROCm 7.2 ships no gfx950 derived-counter section,
/opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots").  Kernels are listed by total GPU time."""
import sqlite3
import sys

CUS, SIMDS = 256, 4
rows = {}
for path in sys.argv[1:]:
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    namecol = "kernel_name" if "kernel_name" in cols else "name"
    idcol = "dispatch_id" if "dispatch_id" in cols else None
    q = "select %s, counter_name, sum(value), count(*) from counters_collection group by 1, 2" % namecol
    for kname, cname, total, n in c.execute(q):
        rows.setdefault(kname or "?", {})[cname] = (total, n)
tab = []
for k, d in rows.items():
    mfma = d.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[0]
    gui, n = d.get("GRBM_GUI_ACTIVE", (0, 0))
    busy = d.get("SQ_BUSY_CU_CYCLES", (0, 0))[0]
    tab.append((gui, k, n, mfma, busy))
tab.sort(reverse=True)
tot_gui = sum(t[0] for t in tab) or 1
tot_mfma = sum(t[3] for t in tab)
print("# MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * %d * %d)" % (CUS, SIMDS))
print("# all kernels of the traced run: %.1f %% of SIMD-cycles had a matrix pipe busy" % (100.0 * tot_mfma / (tot_gui * CUS * SIMDS)))
print("%-90s %7s %9s %9s" % ("kernel", "calls", "time_pct", "mfma_pct"))
for gui, k, n, mfma, busy in tab[:40]:
    print("%-90s %7d %9.2f %9.2f" % (k[:90], n, 100.0 * gui / tot_gui, 100.0 * mfma / (gui * CUS * SIMDS) if gui else 0.0))
