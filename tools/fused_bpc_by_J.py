#!/usr/bin/env python3
"""Fused EM kernel on 1e6 points x component count x workgroups per CU (HGMM_FUSED_BPC): small J leaves most of the
register file unused at the grid that suits J = 800 -- do more waves per SIMD hide the row's reduction chain?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hgmm_amd
ctx = hgmm_amd.Context(0)
N = 1_000_000
X = np.random.RandomState(0).rand(N, 3).astype(np.float32)
ctx.set_points(X)
for J in (64, 100, 200, 400, 512, 800):
    idx = np.random.RandomState(100).choice(N, J, replace=False)
    mu0 = X[idx].copy(); w0 = (np.ones(J) / J).astype(np.float32); cov0 = (0.1 * np.ones((J, 3))).astype(np.float32)
    line = []
    for bpc in ("2", "3", "4", "6", "8"):
        os.environ["HGMM_FUSED_BPC"] = bpc
        ctx.flat_train(10, 0.0, mu0, cov0, w0, "diag", "W")
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); ctx.flat_train(50, 0.0, mu0, cov0, w0, "diag", "W"); ts.append((time.perf_counter() - t0) / 50 * 1e3)
        ctx.profile_reset(); ctx.profile_enable(True)
        ctx.flat_train(20, 0.0, mu0, cov0, w0, "diag", "W")
        ctx.profile_enable(False)
        ms, n = ctx.profile_get("flat_fused")
        line.append("%s wg/CU kernel %.4f iteration %.4f" % (bpc, ms / n, min(ts)))
    print("J=%4d: " % J + " | ".join(line))
