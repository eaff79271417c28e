#!/usr/bin/env python3
"""Where the time of ONE scan pair goes (bench.py --mode pairs: registration_gmmtree of bun000 <-> bun045, L = 3):
wall time of the phases of the reference's call -- GMMTree(source) = upload + buildGMMTree, then registration(target) =
node upload + target upload + the 20-iteration loop -- beside the kernels' hipEvent time.
    python tools/pair_probe.py   ->  profiles/r05/pair_probe.log"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench            # noqa: E402
import hgmm_amd         # noqa: E402
from hgmm_amd.hgmm.hgmm_gpu import GMMTree, n_total_nodes   # noqa: E402

ctx = hgmm_amd.Context(0)
source, pairs = bench.scan_pairs(0, 4)
kw = bench.PAIR_KW
for _ in range(3):
    bench.register_pair(ctx, source, pairs[0][0])
rows = []
for rep in range(8):
    tgt = pairs[rep % 4][0]
    ctx.synchronize()
    t0 = time.perf_counter()
    P = np.ascontiguousarray(source, dtype=np.float64)
    ctx.set_points(P)
    t1 = time.perf_counter()
    T = n_total_nodes(kw["tree_level"])
    idx = np.random.RandomState(72).randint(T, size=T)
    init = P[idx]
    t2 = time.perf_counter()
    pi, mu, cov, leaf, iters, q = ctx.tree_build(kw["tree_level"], kw["ls"], 1e-4, init, kw["sig2"], 1000, want_leaf=False)
    t3 = time.perf_counter()
    ctx.tree_set_nodes(kw["tree_level"], pi, mu, cov)
    t4 = time.perf_counter()
    ctx.tree_set_target(tgt)
    t5 = time.perf_counter()
    rot, t, done, qq, status, _ = ctx.tree_register(np.eye(3), np.zeros(3), 1.0, kw["lambda_c"], 20, 1e-4)
    t6 = time.perf_counter()
    rows.append([t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t6 - t0])
r = np.median(np.array(rows), axis=0) * 1e3
print("median of 8 pairs, ms: set_points(source) %.3f | init draw %.3f | tree_build %.3f (%s level iterations) | set_nodes %.3f | "
      "set_target %.3f | tree_register %.3f (%d iterations, status %d) | total %.3f"
      % (r[0], r[1], r[2], list(iters), r[3], r[4], r[5], done, status, r[6]))
ts = []
for rep in range(8):
    t0 = time.perf_counter()
    bench.register_pair(ctx, source, pairs[rep % 4][0])
    ts.append(time.perf_counter() - t0)
print("the same through GMMTree / registration (the mirror's classes): median %.3f ms" % (np.median(ts) * 1e3))
ctx.profile_reset()
ctx.profile_enable(True)
bench.register_pair(ctx, source, pairs[0][0])
ctx.synchronize()
ctx.profile_enable(False)
for k in ("tree_estep", "tree_loglik", "tree_reg"):
    ms, n = ctx.profile_get(k)
    print("kernels %-12s %.3f ms in %d launches" % (k, ms, n))
ctx.close()
