#!/usr/bin/env python3
"""k-means++ seeding only (C3 frame, k = 800, three repetitions) -- run under rocprofv3 --kernel-trace --stats."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hgmm_amd
n = int(os.environ.get("KMPP_N", "1000000")); k = int(os.environ.get("KMPP_K", "800"))
ctx = hgmm_amd.Context(0)
X = np.random.RandomState(0).rand(n, 3)
ctx.set_points(X - X.mean(0))
rs = np.random.RandomState(1)
trials = 2 + int(np.log(k))
rand = rs.uniform(size=(k - 1, trials))
for rep in range(4):
    t0 = time.perf_counter()
    ids, c = ctx.kmeans_plusplus(k, 12345 % n, rand)
    print("seeding %.2f ms (%.1f us/centre)" % ((time.perf_counter() - t0) * 1e3, (time.perf_counter() - t0) * 1e6 / k))
