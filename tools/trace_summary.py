#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace CSV: per kernel name calls / mean / min / max duration, the idle gap in front
of each launch (start - previous end on the same queue), and optionally the launch sequence of the last build.
    python tools/trace_summary.py <dir> [--seq N]"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
seq = int(sys.argv[sys.argv.index("--seq") + 1]) if "--seq" in sys.argv else 0
rows = []
for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
acc = collections.OrderedDict()
prev_end = None
for s, e, k in rows:
    name = k.split("(")[0].replace("void ", "").replace("hgmm::", "")
    a = acc.setdefault(name, [0, 0, 10 ** 18, 0, 0, 0])
    d = e - s
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    if prev_end is not None:
        g = s - prev_end
        if g < 200000:                      # gaps above 0.2 ms are host pauses between legs, not launch gaps
            a[4] += max(g, 0); a[5] += 1
    prev_end = e
print("%-44s %7s %10s %9s %9s %11s" % ("kernel", "calls", "mean us", "min us", "max us", "gap-before us"))
tot = 0
for name, a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    tot += a[1]
    print("%-44s %7d %10.2f %9.2f %9.2f %11.2f" % (name[:44], a[0], a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3,
                                                      a[4] / max(a[5], 1) / 1e3))
print("total kernel time %.3f ms over %d launches; wall span %.3f ms" % (tot / 1e6, len(rows), (rows[-1][1] - rows[0][0]) / 1e6))
if seq:
    print("last %d launches:" % seq)
    pe = None
    for s, e, k in rows[-seq:]:
        name = k.split("(")[0].replace("void ", "").replace("hgmm::", "")
        print("  %-40s dur %8.2f us  gap %8.2f us" % (name[:40], (e - s) / 1e3, 0 if pe is None else (s - pe) / 1e3))
        pe = e
