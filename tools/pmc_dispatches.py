#!/usr/bin/env python3
"""One counter of a rocprofv3 --pmc run (--output-format csv), per dispatch of the kernels whose name contains the
substring, in dispatch order (16 per line) -- e.g. the VALU instructions of every level-iteration of a batched build.
    python tools/pmc_dispatches.py <dir> <counter> <kernel-substring> [last N]"""
import csv
import glob
import os
import sys

root, counter, sub = sys.argv[1], sys.argv[2], sys.argv[3]
last = int(sys.argv[4]) if len(sys.argv) > 4 else 0
vals = {}
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter or sub not in row["Kernel_Name"]:
                continue
            d = int(row["Dispatch_Id"])
            vals[d] = vals.get(d, 0.0) + float(row["Counter_Value"])
seq = [vals[d] for d in sorted(vals)]
if last:
    seq = seq[-last:]
for i in range(0, len(seq), 16):
    print(" ".join("%.4g" % v for v in seq[i:i + 16]))
print("%d dispatches, sum %.6g" % (len(seq), sum(seq)))
