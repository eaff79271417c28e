#!/usr/bin/env python3
"""HGMM build at scale: per-kernel times (hipEvents) for N points, L levels, fixed iterations."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hgmm_amd
ctx = hgmm_amd.Context(0)
for N, L, iters in ((1_000_000, 3, 6), (1_000_000, 4, 4), (200_000, 4, 6)):
    rs = np.random.RandomState(0)
    cen = rs.rand(300, 3)
    P = cen[rs.randint(300, size=N)] + 0.02 * rs.randn(N, 3)
    T = 8 * (8 ** L - 1) // 7
    idx = rs.randint(T, size=T)
    ctx.set_points(P)
    ctx.tree_build(L, 1e-30, 1e-4, P[idx], 0.002, 2)
    ctx.profile_reset(); ctx.profile_enable(True)
    t0 = time.perf_counter()
    pi, mu, cov, leaf, it, q = ctx.tree_build(L, 1e-30, 1e-4, P[idx], 0.002, iters)
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)
    e_ms, e_n = ctx.profile_get("tree_estep")
    l_ms, l_n = ctx.profile_get("tree_loglik")
    print("N=%d L=%d: build %.1f ms for %d level-iterations | estep avg %.3f ms | loglik avg %.3f ms (levels differ) | live leaves %d"
          % (N, L, dt * 1e3, it.sum(), e_ms / e_n, l_ms / l_n, int((pi[-8 ** L:] > 0).sum())))
