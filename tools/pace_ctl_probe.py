#!/usr/bin/env python3
"""The store pacer's controller (flat_kernels.hip, PaceCtl): started far ABOVE the write path's knee it must walk down to
a rate the memory system takes (HGMM_PACE_START overrides the initial 6700 GB/s for this probe); started at its default
it must stay put.  Prints the target and the kernel time per block of 10 launches, in three call patterns."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hgmm_amd
N, J = 1_000_000, 800
X = np.random.RandomState(0).rand(N, 3).astype(np.float32)
idx = np.random.RandomState(100).choice(N, J, replace=False)
mu0 = X[idx].copy(); w0 = (np.ones(J) / J).astype(np.float32); cov0 = (0.1 * np.ones((J, 3))).astype(np.float32)
for start in (None, "7600"):
    if start:
        os.environ["HGMM_PACE_START"] = start
    ctx = hgmm_amd.Context(0)
    ctx.set_points(X)
    inv, mu, w, cov, lls, _ = ctx.flat_train(10, 0.0, mu0, cov0, w0, "diag", "W")
    lr = ctx.empty((N, J), np.float32)
    print("== initial target %s GB/s" % (start or "default"))
    for blk in range(8):
        ctx.profile_reset(); ctx.profile_enable(True)
        if blk % 2 == 0:
            for _ in range(10):
                ctx.flat_estep(inv, mu, w, "diag", "W", out=lr)
            pat = "blocking"
        else:
            p = (ctx.to_device(inv), ctx.to_device(mu), ctx.to_device(w))
            for _ in range(10):
                ctx.flat_estep(p[0], p[1], p[2], "diag", "W", out=lr, lazy_mean=True)
                ctx.flat_mstep(lr.exp(), "diag", "W", centre_hint=p[1], device_out=True)
            ctx.synchronize()
            pat = "behind m_step"
        ctx.profile_enable(False)
        ms, n = ctx.profile_get("flat_estep")
        print("  block %d (%-13s): kernel %.4f ms   pacer now at %.0f GB/s after %d steps down, %d probes held" % ((blk, pat, ms / n) + ctx.pace_info()), flush=True)
    del lr
    ctx.close()
