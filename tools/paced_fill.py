#!/usr/bin/env python3
"""What the HBM write path takes when a pure store stream is OFFERED at a given rate (util_fill mode 6: the E-step's
store pattern -- 12.8 KB runs dealt round-robin over 2 workgroups per CU -- without its arithmetic, released by the
same StorePacer): the write ceiling the paced E-step is measured against."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hgmm_amd
ctx = hgmm_amd.Context(0)
N, J = 1_000_000, 800
buf = ctx.empty((N, J), np.float32)
nbytes = 4.0 * N * J
rates = [0, 6000, 6400, 6600, 6800, 7000, 7200, 7400, 7600, 7800, 8000]
res = {(r, nt): [] for r in rates for nt in (True, False)}
for rnd in range(3):
    for nt in (True, False):
        for r in rates:
            for _ in range(3):
                ctx.util_fill(buf, float(r), nt, 6, 2)
            ctx.profile_reset(); ctx.profile_enable(True)
            for _ in range(10):
                ctx.util_fill(buf, float(r), nt, 6, 2)
            ctx.profile_enable(False)
            ms, n = ctx.profile_get("util_fill")
            res[(r, nt)].append(ms / n)
print("pure store stream of 3.2 GB, 12.8 KB runs round-robin over 512 workgroups (median of 3 rounds of 10 launches)")
print("%-22s %-34s %s" % ("offered", "non-temporal stores", "plain stores"))
for r in rates:
    a, b = np.median(res[(r, True)]), np.median(res[(r, False)])
    print("%-22s %.4f ms = %5.0f GB/s (%.1f %%)        %.4f ms = %5.0f GB/s (%.1f %%)"
          % ("un-paced" if r == 0 else "%d GB/s" % r, a, nbytes / a / 1e6, nbytes / a / 1e6 / 80, b, nbytes / b / 1e6, nbytes / b / 1e6 / 80))
