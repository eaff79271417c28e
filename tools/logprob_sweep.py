#!/usr/bin/env python3
"""estimate_log_prob (hgmm_flat_log_prob) at C3 size: grid sweep of the four-rows-in-flight raw-table kernel against
the single-row kernel it replaced for this call (interleaved rounds, hipEvent timing)."""
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
inv, mu, w, cov, lls, _ = ctx.flat_train(10, 0.0, mu0, cov0, w0, "diag", "W")
alg = 12 * N + 4 * N * J + 28 * J
cus = ctx.device_info()["compute_units"]
lr = ctx.empty((N, J), np.float32)
cfgs = [("e_step (blocking calls, its own grid policy)", None),
        ("single-row kernel (round 3)", {"HGMM_LOGPROB_SINGLE_ROW": "1"})]
for pace in (0, 1, 2, 3, 4):
    for g in (96, 128, 192, 256, 512):
        cfgs.append(("rows=4 pace=%d grid=%d" % (pace, g), {"HGMM_LOGPROB_GRID": str(g), "HGMM_LOGPROB_PACE": str(pace)}))
res = {k: [] for k, _ in cfgs}
keys = ("HGMM_LOGPROB_SINGLE_ROW", "HGMM_LOGPROB_GRID", "HGMM_LOGPROB_BPC", "HGMM_LOGPROB_LATE", "HGMM_LOGPROB_PACE")
for rnd in range(3):
    for name, env in cfgs:
        for k in keys:
            os.environ.pop(k, None)
        if env is None:
            call = lambda: ctx.flat_estep(inv, mu, w, "diag", "W", out=lr)
        else:
            os.environ.update(env)
            call = lambda: ctx.flat_log_prob(inv, mu, "diag", out=lr)
        for _ in range(3):
            call()
        ctx.profile_reset(); ctx.profile_enable(True)
        for _ in range(12):
            call()
        ctx.profile_enable(False)
        ms, n = ctx.profile_get("flat_estep")
        res[name].append(ms / n)
print("estimate_log_prob, N = %d, J = %d, %d CUs; algorithmic bytes %.4f GB" % (N, J, cus, alg / 1e9))
for name, _ in cfgs:
    v = np.array(res[name])
    print("%-46s median %.4f ms  (%.0f GB/s, %.1f%% of 8 TB/s)   rounds %s"
          % (name, np.median(v), alg / np.median(v) / 1e6, alg / np.median(v) / 1e6 / 80, np.round(v, 4)))
