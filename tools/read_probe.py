#!/usr/bin/env python3
"""What the HBM READ path delivers for the M-step's access pattern without its arithmetic (util_fill mode 7): every wave
reads 3200-byte rows (16-byte loads), 1 - 3 rows requested ahead, rows contiguous per wave or dealt round-robin,
1 - 4 workgroups per CU, plain or non-temporal loads.  The ceiling flat_mstep_kernel (0.51 - 0.53 ms) is up against."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hgmm_amd
ctx = hgmm_amd.Context(0)
N, J = 1_000_000, 800
buf = ctx.empty((N, J), np.float32)
ctx.util_fill(buf, 1.0, False, 0, 1)
nbytes = 4.0 * N * J
rows = []
for nt in (False, True):
    for bpc in (1, 2, 4, 8):
        for depth in (1, 2, 3):
            for rr in (0, 1):
                rows.append((nt, bpc, depth, rr))
res = {r: [] for r in rows}
for rnd in range(3):
    for r in rows:
        nt, bpc, depth, rr = r
        for _ in range(2):
            ctx.util_fill(buf, float(10 * depth + rr), nt, 7, bpc)
        ctx.profile_reset(); ctx.profile_enable(True)
        for _ in range(8):
            ctx.util_fill(buf, float(10 * depth + rr), nt, 7, bpc)
        ctx.profile_enable(False)
        ms, n = ctx.profile_get("util_fill")
        res[r].append(ms / n)
print("pure read stream of 3.2 GB as 3200-byte rows (median of 3 rounds of 8 launches)")
for r in rows:
    t = float(np.median(res[r]))
    print("%-13s %d wg/CU  %d row(s) ahead  %-12s %.4f ms = %5.0f GB/s (%.1f %%)"
          % ("non-temporal" if r[0] else "plain loads", r[1], r[2], "round-robin" if r[3] else "contiguous", t, nbytes / t / 1e6, nbytes / t / 1e6 / 80))
