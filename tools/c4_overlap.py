#!/usr/bin/env python3
"""C4 (bun000, L = 4) build time with and without the E-step of iteration e + 1 riding in iteration e's log-likelihood
launch (HGMM_TREE_OVERLAP, tree_ll_estep_kernel); both forms must return the same tree bit for bit."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hgmm_amd
ctx = hgmm_amd.Context(0)
P = np.load(os.path.join(ROOT, "tests", "golden", "bun000_xyz.npy")).astype(np.float64)
L, T = 4, 4680
idx = np.random.RandomState(72).randint(T, size=T)
ctx.set_points(P)
out = {}
res = {"0": [], "1": []}
for rnd in range(4):
    for ov in ("0", "1"):
        os.environ["HGMM_TREE_OVERLAP"] = ov
        ctx.tree_build(L, 80.0, 1e-4, P[idx], 0.00034, 1000)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            r = ctx.tree_build(L, 80.0, 1e-4, P[idx], 0.00034, 1000, want_leaf=False)
            ts.append(time.perf_counter() - t0)
        res[ov].append(float(np.median(ts)) * 1e3)
        out[ov] = ctx.tree_build(L, 80.0, 1e-4, P[idx], 0.00034, 1000)
for ov in ("0", "1"):
    print("HGMM_TREE_OVERLAP=%s: build %.3f ms (rounds %s), level iterations %s" % (ov, np.median(res[ov]), np.round(res[ov], 3), out[ov][4]))
a, b = out["0"], out["1"]
same = all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(a, b))
print("trees, leaf assignment, iteration counts and q traces bitwise equal:", same)
