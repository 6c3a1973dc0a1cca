"""Per-step kernel table from a rocprofv3 kernel-trace CSV: only the steady-state steps are counted (the process also runs
MIOpen's find phase, whose candidate kernels -- naive_conv_* among them -- would otherwise dominate the totals).

    python tools/trace_window.py <kernel_trace.csv> [steps=5] [marker=roi_align_bwd_tiles]

A kernel that runs a fixed number of times per step (default: the RoIAlign tile backward, two launches per step: box and
mask head) delimits the steps; the window spans `steps` whole steps ending at the last marker launch."""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    marker = sys.argv[3] if len(sys.argv) > 3 else "roi_align_bwd_tiles"
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [s for s, _, n in rows if marker in n]
    per_step = 2
    assert len(marks) >= per_step * (steps + 1), "not enough steps traced (%d marker launches)" % len(marks)
    t1 = marks[-1]
    t0 = marks[-1 - per_step * steps]
    agg = defaultdict(lambda: [0, 0])
    busy = 0
    for s, e, n in rows:
        if t0 <= s < t1:
            agg[n][0] += 1
            agg[n][1] += e - s
            busy += e - s
    wall = (t1 - t0) / steps
    print("window: %d steps, %.3f ms wall per step, %.3f ms of kernel time per step (%.1f %% of wall)" %
          (steps, wall / 1e6, busy / steps / 1e6, 100.0 * busy / steps / wall))
    print("%7s %9s %11s %11s  %s" % ("share", "calls/st", "us/step", "avg us", "kernel"))
    groups = defaultdict(float)
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        key = ("MIOpen Winograd (no MFMA)" if "Sp3AsmConv" in n else
               "MIOpen implicit GEMM / CK conv (MFMA)" if ("igemm" in n or "ck::" in n or "_ZN2ck" in n) else
               "hipBLASLt / Tensile GEMM (MFMA)" if n.startswith("Cijk") else
               "this library (mi_*)" if ("mi::" in n or "affine_" in n or "(anonymous namespace)::" in n and "at::" not in n) else
               "MIOpen layout / im2col helpers" if ("transpose" in n or "Im2d2Col" in n or "Col2Im" in n or "SubTensor" in n) else
               "PyTorch element-wise / reduce / index")
        groups[key] += t
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:120]:
        print("%6.2f%% %9.1f %11.1f %11.1f  %s" % (100.0 * t / busy, c / steps, t / steps / 1e3, t / c / 1e3, n[:140]))
    print("\ngroups (share of kernel time):")
    for k, t in sorted(groups.items(), key=lambda kv: -kv[1]):
        print("%6.2f%% %9.3f ms/step  %s" % (100.0 * t / busy, t / steps / 1e6, k))


if __name__ == "__main__":
    main()
