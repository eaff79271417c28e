#!/usr/bin/env python3
"""Per-level kernel means of the LAST C4 build in a rocprofv3 kernel trace (levels are separated by tree_chunks_kernel):
    rocprofv3 --kernel-trace --output-format csv -d <dir> -o kt -- python tools/c4prof.py c4 3;  python tools/c4_levels.py <dir>"""
import collections, csv, glob, os, sys
rows = []
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hgmm::", "")))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("tree_init_nodes_kernel")]
last = rows[starts[-1]:]
level, acc = -1, collections.OrderedDict()
t_first = last[0][0]
for s, e, k in last:
    if k.startswith("tree_chunks_kernel"):
        level += 1
    a = acc.setdefault((level, k), [0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3
for (lv, k), (n, us) in acc.items():
    print("level %2d  %-34s x%3d  mean %6.2f us  total %7.1f us" % (lv, k, n, us / n, us))
print("span of the build's kernels: %.1f us" % ((last[-1][1] - t_first) / 1e3))
