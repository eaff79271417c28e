#!/usr/bin/env python3
"""HBM pure-write ceiling: sweep of store patterns (mode, grid size, temporal hint)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hgmm_amd
ctx = hgmm_amd.Context(0)
n = 800_000_000
buf = ctx.empty((n,), np.float32)
def run(label, value, nt, mode, gm):
    ctx.util_fill(buf, value, nt, mode, gm)
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(8):
        ctx.util_fill(buf, value, nt, mode, gm)
    ctx.profile_enable(False)
    ms, k = ctx.profile_get("util_fill")
    print("%s : %.4f ms  %.0f GB/s" % (label, ms / k, 4 * n / (ms / k * 1e-3) / 1e9))
for rnd in range(2):
    for gm in (1, 2, 4):
        for nt in (False, True):
            run("round %d mode0 grid %dxCU x256thr nt=%d" % (rnd, gm, nt), 1.0, nt, 0, gm)
    for gm in (1, 2, 4):
        for rows in (1, 4, 8, 16, 32):
            chunk = rows * 200            # rows of 800 floats = 200 float4
            for nt in (False, True):
                run("round %d mode3 writers/CU=%d chunk=%2d rows (%5.1f KB) nt=%d" % (rnd, gm, rows, rows * 3.2, nt), float(chunk), nt, 3, gm)
