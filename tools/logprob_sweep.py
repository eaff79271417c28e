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
cfgs = [("single-row kernel (round 3)", {"HGMM_LOGPROB_SINGLE_ROW": "1"})]
for g in (192, 256, 512, 768, 1024, 1280, 1536, 2048, 3072, 4096):
    cfgs.append(("rows=4 grid=%d" % g, {"HGMM_LOGPROB_GRID": str(g)}))
res = {k: [] for k, _ in cfgs}
keys = ("HGMM_LOGPROB_SINGLE_ROW", "HGMM_LOGPROB_GRID", "HGMM_LOGPROB_BPC")
out = None
for rnd in range(3):
    for name, env in cfgs:
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(env)
        for _ in range(3):
            out = None
            out = ctx.flat_log_prob(inv, mu, "diag")
        ctx.profile_reset(); ctx.profile_enable(True)
        for _ in range(12):
            out = None
            out = ctx.flat_log_prob(inv, mu, "diag")
        ctx.profile_enable(False)
        ms, n = ctx.profile_get("flat_estep")
        res[name].append(ms / n)
print("estimate_log_prob, N = %d, J = %d, %d CUs; algorithmic bytes %.4f GB" % (N, J, cus, alg / 1e9))
for name, _ in cfgs:
    v = np.array(res[name])
    print("%-28s median %.4f ms  (%.0f GB/s, %.1f%% of 8 TB/s)   rounds %s"
          % (name, np.median(v), alg / np.median(v) / 1e6, alg / np.median(v) / 1e6 / 80, np.round(v, 4)))
