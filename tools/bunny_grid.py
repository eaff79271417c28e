#!/usr/bin/env python3
"""bun000 fits (20 iterations) x workgroups of the fused kernel (HGMM_FUSED_GRID): small clouds give a wave a handful of
rows under the large-cloud grid rule, and every workgroup costs a partial the reduction has to read."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hgmm_amd
ctx = hgmm_amd.Context(0)
X = np.load(os.path.join(ROOT, "tests", "golden", "bun000_xyz.npy")).astype(np.float32)
ctx.set_points(X)
for J in (100, 800):
    idx = np.random.RandomState(100).choice(len(X), J, replace=False)
    mu0 = X[idx].copy(); w0 = (np.ones(J) / J).astype(np.float32); cov0 = (0.1 * np.ones((J, 3))).astype(np.float32)
    for grid in (0, 512, 384, 256, 192, 128, 96, 64, 32):
        os.environ["HGMM_FUSED_GRID"] = str(grid)
        ctx.flat_train(20, 0.0, mu0, cov0, w0, "diag", "W")
        ts = []
        for _ in range(15):
            t0 = time.perf_counter()
            ctx.flat_train(20, 0.0, mu0, cov0, w0, "diag", "W")
            ts.append(time.perf_counter() - t0)
        ctx.profile_reset(); ctx.profile_enable(True)
        ctx.flat_train(20, 0.0, mu0, cov0, w0, "diag", "W")
        ctx.profile_enable(False)
        ms, n = ctx.profile_get("flat_fused")
        print("J=%d grid %4s: 20 iterations %.3f ms (%.0f it/s), fused kernel %.2f us" % (J, grid or "rule", np.median(ts) * 1e3, 20 / np.median(ts), ms / n * 1e3))
