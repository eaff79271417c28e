#!/usr/bin/env python3
"""A forest build of B bunny scans with a FIXED number of iterations per level (ls = 0: the stop rule never fires), for
timing experiments in which the arithmetic is deliberately altered: wall time of hgmm_tree_build_batch.
    python tools/forest_fixed_probe.py [B] [iterations per level] [levels] [--f32]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import hgmm_amd  # noqa: E402
from hgmm_amd.hgmm.hgmm_gpu import n_total_nodes  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(args[0]) if args else 32
iters = int(args[1]) if len(args) > 1 else 20
ctx = hgmm_amd.Context(0)
if "--f32" in sys.argv:
    ctx.tree_set_precision(np.float32)
source, _ = bench.scan_pairs(0, 1)
L = int(args[2]) if len(args) > 2 else 3
T = n_total_nodes(L)
idx = np.random.RandomState(72).randint(T, size=T)
arrs = ctx.set_points_batch([source] * B)
init = np.stack([a[idx] for a in arrs])
ts = []
for rep in range(6):
    t0 = time.perf_counter()
    _, it, _ = ctx.tree_build_batch([len(a) for a in arrs], L, 0.0, 1e-4, init, 0.004, iters, want_tables=False)
    ts.append(time.perf_counter() - t0)
print("B = %d, %d iterations per level (%s): build %.3f ms (median of 5), iterations %s"
      % (B, iters, "float32 pdfs" if "--f32" in sys.argv else "float64", 1e3 * float(np.median(ts[1:])), it[0].tolist()))
ctx.close()
