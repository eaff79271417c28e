#!/usr/bin/env python3
"""The store pacer's controller over a long mixed sequence of E-step launches (blocking calls, an unwaited-for stream, a
caller's e_step -> m_step loop, in turn): mean kernel time per 120 launches and where the rate ends, for a fixed rate,
the controller that only backs off, and the controller that also probes upwards."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hgmm_amd
N, J = 1_000_000, 800
X = np.random.RandomState(0).rand(N, 3).astype(np.float32)
idx = np.random.RandomState(100).choice(N, J, replace=False)
mu0 = X[idx].copy(); w0 = (np.ones(J) / J).astype(np.float32); cov0 = (0.1 * np.ones((J, 3))).astype(np.float32)
scen = [("fixed 6600 GB/s", {"HGMM_ESTEP_TARGET_GBS": "6600"}),
        ("controller, backs off only", {"HGMM_ESTEP_PROBE": "0"}),
        ("controller, probes upwards too", {})]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for name, env in scen:
    for k in ("HGMM_ESTEP_TARGET_GBS", "HGMM_ESTEP_PROBE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ctx = hgmm_amd.Context(0)
    ctx.set_points(X)
    inv, mu, w, cov, lls, _ = ctx.flat_train(10, 0.0, mu0, cov0, w0, "diag", "W")
    lr = ctx.empty((N, J), np.float32)
    for _ in range(3):
        ctx.flat_estep(inv, mu, w, out=lr)
    line = []
    tot_ms, tot_n = 0.0, 0
    for rep in range(reps):
        ctx.profile_reset(); ctx.profile_enable(True)
        for _ in range(40):
            ctx.flat_estep(inv, mu, w, out=lr)
        for _ in range(40):
            m = ctx.flat_estep(inv, mu, w, out=lr, lazy_mean=True)[0]
            float(m)                                                  # (a caller that looks at the mean: one launch ahead at most)
        p = (ctx.to_device(inv), ctx.to_device(mu), ctx.to_device(w))
        for _ in range(40):
            m = ctx.flat_estep(p[0], p[1], p[2], out=lr, lazy_mean=True)[0]
            ctx.flat_mstep(lr.exp(), "diag", "W", centre_hint=p[1], device_out=True)
            float(m)
        ctx.synchronize()
        ctx.profile_enable(False)
        ms, n = ctx.profile_get("flat_estep")
        tot_ms += ms; tot_n += n
        t, st, up = ctx.pace_info()
        line.append("%.4f@%d" % (ms / n, t))
    t, st, up = ctx.pace_info()
    print("%-32s mean %.4f ms over %d launches; per 120: %s; steps down %d, probes held %d"
          % (name, tot_ms / tot_n, tot_n, " ".join(line), st, up), flush=True)
    del lr
    ctx.close()
