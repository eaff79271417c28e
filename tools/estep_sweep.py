#!/usr/bin/env python3
"""A/B sweep of the materialising E-step launch shapes on the GPU box (interleaved, hipEvent timing)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hgmm_amd
ctx = hgmm_amd.Context(0)
N, J = 1_000_000, 800
X = np.random.RandomState(0).rand(N, 3).astype(np.float32)
idx = np.random.RandomState(100).choice(N, J, replace=False)
mu0 = X[idx].copy(); w0 = (np.ones(J) / J).astype(np.float32); cov0 = (0.1 * np.ones((J, 3))).astype(np.float32)
ctx.set_points(X)
inv, mu, w, cov, lls, _ = ctx.flat_train(30, 0.0, mu0, cov0, w0, "diag", "W")
lr = ctx.empty((N, J), np.float32)
alg = 12 * N + 4 * N * J + 4 * N + 28 * J
cfgs = [("rows=4 bpc=1", {"HGMM_ESTEP_ROWS": "4", "HGMM_ESTEP_BPC": "1"}),
        ("rows=1 bpc=2", {"HGMM_ESTEP_ROWS": "1", "HGMM_ESTEP1_BPC": "2"}),
        ("rows=1 bpc=1", {"HGMM_ESTEP_ROWS": "1", "HGMM_ESTEP1_BPC": "1"}),
        ("rows=1 bpc=3", {"HGMM_ESTEP_ROWS": "1", "HGMM_ESTEP1_BPC": "3"}),
        ("rows=1 bpc=2 rr", {"HGMM_ESTEP_ROWS": "1", "HGMM_ESTEP1_BPC": "2", "HGMM_ESTEP_RR": "1"}),
        ("rows=1 bpc=1 rr", {"HGMM_ESTEP_ROWS": "1", "HGMM_ESTEP1_BPC": "1", "HGMM_ESTEP_RR": "1"})]
res = {k: [] for k, _ in cfgs}
for rnd in range(4):
    for name, env in cfgs:
        for k in ("HGMM_ESTEP_ROWS", "HGMM_ESTEP_BPC", "HGMM_ESTEP1_BPC", "HGMM_ESTEP_RR"):
            os.environ.pop(k, None)
        os.environ.update(env)
        for _ in range(3):
            ctx.flat_estep(inv, mu, w, "diag", "W", out=lr)
        ctx.profile_reset(); ctx.profile_enable(True)
        for _ in range(15):
            ctx.flat_estep(inv, mu, w, "diag", "W", out=lr)
        ctx.profile_enable(False)
        ms, n = ctx.profile_get("flat_estep")
        res[name].append(ms / n)
for name, _ in cfgs:
    v = np.array(res[name])
    print("%-18s median %.4f ms  (%.0f GB/s, %.1f%% of 8 TB/s)   runs %s" % (name, np.median(v), alg / np.median(v) / 1e6, alg / np.median(v) / 1e6 / 80, np.round(v, 4)))
