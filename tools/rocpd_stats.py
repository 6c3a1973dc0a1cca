#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max.

    python tools/rocpd_stats.py gpurun_out/<dir>/<name>_results.db [> profiles/<name>.txt]

Equivalent to the kernel section of `rocprofv3 --stats` CSV output; used to turn the scratch
databases under gpurun_out/ into the small text summaries committed under profiles/.
"""
import sqlite3
import sys


def main(path, pmc=False):
    c = sqlite3.connect(path)
    rows = c.execute(
        "select k.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
        "max(d.end - d.start), max(k.arch_vgpr_count), max(k.sgpr_count), max(d.group_segment_size), "
        "max(d.workgroup_size_x), max(d.grid_size_x) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol k on d.kernel_id = k.id "
        "group by k.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("%-72s %7s %12s %10s %10s %10s %6s %5s %5s %7s %6s %9s" % (
        "kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds_B", "wg", "grid"))
    for name, calls, tot, avg, mn, mx, vg, sg, lds, wg, grid in rows:
        short = name if len(name) <= 72 else name[:69] + "..."
        print("%-72s %7d %12.1f %10.2f %10.2f %10.2f %6.2f %5s %5s %7s %6s %9s" % (
            short, calls, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total, vg, sg, lds, wg, grid))
    if pmc:
        try:
            rows = c.execute("select * from counters_collection limit 0")
            cols = [d[0] for d in rows.description]
            print("\n# counters_collection columns:", cols)
            q = ("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection "
                 "group by kernel_name, counter_name order by kernel_name, counter_name")
            for r in c.execute(q):
                print("%-60s %-28s n=%-6d avg=%-16.1f sum=%.1f" % ((r[0] or "")[:60], r[1], r[2], r[3], r[4]))
        except Exception as e:  # pragma: no cover
            print("# no counter data:", e)


if __name__ == "__main__":
    main(sys.argv[1], pmc="--pmc" in sys.argv)
