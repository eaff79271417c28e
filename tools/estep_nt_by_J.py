#!/usr/bin/env python3
"""The materialising E-step with plain / non-temporal stores x row length (rows that are not whole 128-byte lines share
lines with their neighbours)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hgmm_amd
ctx = hgmm_amd.Context(0)
N = 1_000_000
X = np.random.RandomState(0).rand(N, 3).astype(np.float32)
ctx.set_points(X)
for J in (800, 100, 72, 200, 37, 513, 1000):
    idx = np.random.RandomState(100).choice(N, J, replace=False)
    mu = X[idx].copy(); w = (np.ones(J) / J).astype(np.float32); inv = (np.ones((J, 3)) / np.sqrt(0.05)).astype(np.float32)
    lr = ctx.empty((N, J), np.float32)
    out = []
    for nt in ("0", "1"):
        os.environ["HGMM_ESTEP_NT"] = nt
        r = []
        for rnd in range(3):
            for _ in range(3): ctx.flat_estep(inv, mu, w, "diag", "W", out=lr)
            ctx.profile_reset(); ctx.profile_enable(True)
            for _ in range(10): ctx.flat_estep(inv, mu, w, "diag", "W", out=lr)
            ctx.profile_enable(False)
            ms, n = ctx.profile_get("flat_estep"); r.append(ms / n)
        out.append(float(np.median(r)))
    gb = (4.0 * N * J + 16.0 * N) / 1e6
    print("J=%4d (%5d-byte rows): plain stores %.4f ms = %5.0f GB/s | non-temporal %.4f ms = %5.0f GB/s" % (J, 4 * J, out[0], gb / out[0], out[1], gb / out[1]))
    del lr
