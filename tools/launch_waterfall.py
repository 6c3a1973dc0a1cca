#!/usr/bin/env python3
"""Waterfall of the launches of one C-ABI call from a rocprofv3 --kernel-trace CSV: for every repetition of the kernel
sequence (a call = the run of kernels up to and including LAST), start offset and duration of each launch relative to the
start of the call's first kernel, averaged over the calls after the warm-up; and the wall time of a call = last end - first
start, beside the period between consecutive calls (what HIP events around back-to-back calls measure).

usage: python tools/launch_waterfall.py <kernel_trace.csv> <substring of the LAST kernel of a call> [skip_calls]"""
import csv
import sys
from collections import defaultdict


def short(name):
    return name.replace("void ", "").replace("mi::(anonymous namespace)::", "").split("(")[0].split("<")[0]


def main():
    path, last = sys.argv[1], sys.argv[2]
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    rows = [r for r in csv.DictReader(open(path)) if "roi_align" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    calls, cur = [], []
    for r in rows:
        cur.append((short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        if last in r["Kernel_Name"]:
            calls.append(cur)
            cur = []
    calls = calls[skip:]
    shape = max(set(tuple(k for k, _, _ in c) for c in calls), key=lambda s: sum(1 for c in calls if tuple(k for k, _, _ in c) == s))
    calls = [c for c in calls if tuple(k for k, _, _ in c) == shape]
    acc = defaultdict(lambda: [0.0, 0.0])
    wall = 0.0
    for c in calls:
        t0 = c[0][1]
        for i, (k, s, e) in enumerate(c):
            acc[i][0] += (s - t0) / 1e3
            acc[i][1] += (e - s) / 1e3
        wall += (c[-1][2] - t0) / 1e3
    n = len(calls)
    period = (calls[-1][0][1] - calls[0][0][1]) / 1e3 / max(n - 1, 1)
    print("%d calls of %d launches" % (n, len(shape)))
    prev_end = 0.0
    for i, k in enumerate(shape):
        s, d = acc[i][0] / n, acc[i][1] / n
        print("  %-28s starts %7.2f us  runs %7.2f us  ends %7.2f us   (gap to the previous end %+6.2f us)" % (k, s, d, s + d, s - prev_end))
        prev_end = s + d
    print("  first start -> last end %.2f us; period of back-to-back calls %.2f us" % (wall / n, period))


if __name__ == "__main__":
    main()
