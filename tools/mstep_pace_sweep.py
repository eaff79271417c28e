#!/usr/bin/env python3
"""m_step(X, exp(log_resp)) at C3 size x workgroups per CU (the read-side pacing experiment of round 4 -- a StorePacer
before every row load, HGMM_MSTEP_TARGET_GBS -- lost at every target and was removed: profiles/r04/mstep_pace.log)."""
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
lr = ctx.empty((N, J), np.float32)
ctx.flat_estep(inv, mu, w, "diag", "W", out=lr)
alg = 4 * N * J + 12 * N
cfgs = []
for nt in ("0", "1"):
    for rr in ("0", "1"):
        for bpc in ("2", "3"):
            cfgs.append(("%s loads, rows %s, %s wg/CU" % ("non-temporal" if nt == "1" else "plain", "round-robin" if rr == "1" else "contiguous", bpc),
                         {"HGMM_MSTEP_NT": nt, "HGMM_MSTEP_RR": rr, "HGMM_MSTEP_BPC": bpc}))
res = {c: [] for c, _ in cfgs}
dmu = ctx.to_device(mu)
for rnd in range(3):
    for name, env in cfgs:
        os.environ.update(env)
        for _ in range(3):
            ctx.flat_mstep(lr.exp(), "diag", "W", centre_hint=dmu, device_out=True)
        ctx.profile_reset(); ctx.profile_enable(True)
        for _ in range(10):
            ctx.flat_mstep(lr.exp(), "diag", "W", centre_hint=dmu, device_out=True)
        ctx.profile_enable(False)
        ms, n = ctx.profile_get("flat_mstep")
        res[name].append(ms / n)
for name, _ in cfgs:
    v = np.array(res[name])
    print("%-46s median %.4f ms  (%.0f GB/s, %.1f%% of 8 TB/s)  rounds %s" % (name, np.median(v), alg / np.median(v) / 1e6, alg / np.median(v) / 1e6 / 80, np.round(v, 4)))
