#!/usr/bin/env python3
"""Per-grid breakdown of the materialising E-step's launches in a rocprofv3 kernel trace of `python bench.py`.

    python tools/estep_patterns.py gpurun_out/prof_r03/kt/kt_kernel_trace.csv

The kernel runs in two call patterns inside one bench run, told apart by their grids (flat_kernels.hip,
estep_rows_grid): 192 workgroups = blocking hgmm_flat_estep calls (the `roofline` leg: steady + cold figures; the host
reads the mean after every launch), 512 workgroups = launches behind an M-step or behind an E-step nobody waited for
(`roofline.unsynchronised_stream_*`, both loops of `materialised_iteration`).  rocprofv3's one average per kernel
symbol mixes the two; `roofline.avg_launch_ms` is the first group's."""
import collections
import csv
import statistics
import sys

rows = csv.DictReader(open(sys.argv[1]))
groups = collections.defaultdict(list)
for r in rows:
    if "flat_estep_rows_pk_kernel" in r["Kernel_Name"]:
        wg = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
        groups[wg].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-12s %8s %10s %10s %10s" % ("workgroups", "launches", "mean us", "min us", "max us"))
tot = []
for wg in sorted(groups):
    v = groups[wg]
    tot += v
    print("%-12d %8d %10.1f %10.1f %10.1f" % (wg, len(v), statistics.mean(v), min(v), max(v)))
if tot:
    print("%-12s %8d %10.1f %10.1f %10.1f" % ("all", len(tot), statistics.mean(tot), min(tot), max(tot)))
