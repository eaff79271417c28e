#!/usr/bin/env python3
"""The HGMM builds bench.py times, alone, for a kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d <dir> -o kt -- python tools/c4prof.py [c4|tree1m|both] [reps]
then  python tools/trace_summary.py <dir>  (per-kernel durations in launch order, gaps between launches)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hgmm_amd  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "both"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
# C4PROF_LEAF=1: also un-sort and download the N-long leaf assignment (bench.py's build_ms is the node tables only,
# what the reference's buildGMMTree returns)
LEAF = os.environ.get("C4PROF_LEAF", "0") == "1"
ctx = hgmm_amd.Context(0)
L, T = 4, 4680
if what in ("c4", "both"):
    P = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                             "bun000_xyz.npy")).astype(np.float64)
    idx = np.random.RandomState(72).randint(T, size=T)
    ctx.set_points(P)
    ctx.tree_build(L, 80.0, 1e-4, P[idx], 0.00034, 1000)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = ctx.tree_build(L, 80.0, 1e-4, P[idx], 0.00034, 1000, want_leaf=LEAF)
        ts.append(time.perf_counter() - t0)
    print("C4 build ms", [round(t * 1e3, 3) for t in ts], "iterations", list(out[4]), flush=True)
if what in ("tree1m", "both"):
    P = np.random.RandomState(0).rand(1_000_000, 3).astype(np.float32).astype(np.float64)
    idx = np.random.RandomState(72).randint(len(P), size=T)
    ctx.set_points(P)
    ctx.tree_build(L, 80.0, 1e-4, P[idx], 0.01, 4)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = ctx.tree_build(L, 80.0, 1e-4, P[idx], 0.01, 4, want_leaf=LEAF)
        ts.append(time.perf_counter() - t0)
    pi = out[0]
    print("tree_1M build ms", [round(t * 1e3, 3) for t in ts], "dead nodes per level",
          [int((pi[8 * (8 ** l - 1) // 7: 8 * (8 ** (l + 1) - 1) // 7] == 0).sum()) for l in range(L)], flush=True)
ctx.close()
