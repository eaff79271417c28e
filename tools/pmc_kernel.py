#!/usr/bin/env python3
"""Mean per launch of every counter of a rocprofv3 --pmc run (--output-format csv), per kernel.
    python tools/pmc_kernel.py <dir> [kernel-substring]"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            if sub and sub not in k:
                continue
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
for k, cs in sorted(acc.items()):
    print(k[:110])
    for c, (tot, n) in sorted(cs.items()):
        print("   %-24s %14.6g  (mean of %d launches)" % (c, tot / n, n))
