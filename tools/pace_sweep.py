#!/usr/bin/env python3
"""Store pacing (StorePacer, csrc/flat_kernels.hip) of the materialising E-step and of estimate_log_prob at C3 size:
target rate x workgroups per CU, in the three call patterns bench.py times -- blocking calls, an unwaited-for stream,
behind an M-step (a caller's e_step -> m_step loop).  hipEvent time of the kernel, interleaved rounds."""
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
alg = 12 * N + 4 * N * J + 4 * N + 28 * J
targets = [int(t) for t in (sys.argv[1].split(",") if len(sys.argv) > 1 else "5600,5800,6000,6200,6400".split(","))]
bpcs = [int(b) for b in (sys.argv[2].split(",") if len(sys.argv) > 2 else "1,2,4".split(","))]
cfgs = [("round-3 grid policy, no pacing", {"HGMM_ESTEP_TARGET_GBS": "0", "HGMM_LOGPROB_TARGET_GBS": "0"})]
for t in targets:
    for b in bpcs:
        cfgs.append(("target %d GB/s, %d wg/CU" % (t, b), {"HGMM_ESTEP_TARGET_GBS": str(t), "HGMM_ESTEP_BPC": str(b),
                                                          "HGMM_LOGPROB_TARGET_GBS": str(t), "HGMM_LOGPROB_BPC": str(b)}))
keys = ("HGMM_ESTEP_TARGET_GBS", "HGMM_ESTEP_BPC", "HGMM_LOGPROB_TARGET_GBS", "HGMM_LOGPROB_BPC")


def blocking():
    for _ in range(12):
        ctx.flat_estep(inv, mu, w, "diag", "W", out=lr)


def stream():
    for _ in range(12):
        ctx.flat_estep(inv, mu, w, "diag", "W", out=lr, lazy_mean=True)
    ctx.synchronize()


def behind_mstep():
    p = (ctx.to_device(inv), ctx.to_device(mu), ctx.to_device(w))
    for _ in range(8):
        ctx.flat_estep(p[0], p[1], p[2], "diag", "W", out=lr, lazy_mean=True)
        ctx.flat_mstep(lr.exp(), "diag", "W", centre_hint=p[1], device_out=True)
    ctx.synchronize()


def logprob():
    for _ in range(12):
        ctx.flat_log_prob(inv, mu, "diag", out=lr)


def with_argmax():
    for _ in range(12):
        ctx.flat_estep(inv, mu, w, "diag", "W", out=lr, want_argmax=True, want_lpn=True)


patterns = (("blocking", blocking), ("stream", stream), ("behind m_step", behind_mstep), ("log_prob", logprob),
            ("row-max loop", with_argmax))
res = {(c, p): [] for c, _ in cfgs for p, _ in patterns}
for rnd in range(3):
    for name, env in cfgs:
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(env)
        for pname, fn in patterns:
            fn()
            ctx.profile_reset(); ctx.profile_enable(True)
            fn()
            ctx.profile_enable(False)
            ms, n = ctx.profile_get("flat_estep")
            res[(name, pname)].append(ms / n)
print("materialising E-step / estimate_log_prob, N = %d, J = %d: kernel ms (median of 3 rounds) per call pattern" % (N, J))
print("%-34s %s   mean of the three e_step patterns" % ("", "  ".join("%-14s" % p for p, _ in patterns)))
for name, _ in cfgs:
    med = [float(np.median(res[(name, p)])) for p, _ in patterns]
    m3 = float(np.mean(med[:3]))
    print("%-34s %s   %.4f ms = %.1f %% of 8 TB/s" % (name, "  ".join("%-14.4f" % v for v in med), m3, alg / m3 / 1e6 / 80))
