#!/usr/bin/env python3
"""How busy the GPU was over a rocprofv3 --kernel-trace: union of the kernels' intervals over the span they cover (a
launch-gap / host-bound measure), the mean number of kernels in flight, and the idle gaps by length.
    python tools/gpu_busy.py <dir> [skip-first-fraction]"""
import csv
import glob
import os
import sys

rows = []
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
t_lo = rows[0][0] + skip * (rows[-1][1] - rows[0][0])
rows = [r for r in rows if r[0] >= t_lo]
span = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
gaps = []
for s, e in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e in rows)
print("launches %d  span %.1f ms  busy (union) %.1f ms = %.1f %%  sum of durations %.1f ms = %.2f kernels in flight on average"
      % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, tot / 1e6, tot / span))
for lo, hi in ((0, 5e3), (5e3, 2e4), (2e4, 1e5), (1e5, 1e6), (1e6, 1e12)):
    g = [x for x in gaps if lo <= x < hi]
    print("  idle gaps %7.0f .. %-9.0f us: %6d, %.2f ms in all" % (lo / 1e3, hi / 1e3, len(g), sum(g) / 1e6))
