#!/usr/bin/env python3
"""Per launch of the LAST batched build in a rocprofv3 --kernel-trace directory: the durations of the forest kernels in
launch order (one line per level-iteration), so that the cost of a level can be read off.
    python tools/forest_levels.py <dir>"""
import csv
import glob
import os
import sys

rows = []
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
names = [k.split("(")[0].replace("void ", "").replace("hgmm::", "") for _, _, k in rows]
# the last build: from the last forest_prep_kernel to the first forest_reg_estep behind it
last = max(i for i, n in enumerate(names) if n.startswith("forest_prep_kernel"))
end = next((i for i in range(last, len(rows)) if names[i].startswith("forest_reg_estep")), len(rows))
t0 = rows[last][0]
it = 0
for i in range(last, end):
    s, e, _ = rows[i]
    n = names[i]
    print("%4d %-34s start %9.1f us  dur %8.2f us  gap %7.2f us" % (i - last, n[:34], (s - t0) / 1e3, (e - s) / 1e3,
                                                                  (s - rows[i - 1][1]) / 1e3))
print("build span %.3f ms" % ((rows[end - 1][1] - t0) / 1e6))
