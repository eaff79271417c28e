import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import hgmm_amd, bench
ctx = hgmm_amd.Context(0)
X = bench.synth_frame(0); mu0, w0, cov0 = bench.init_params(X)
ctx.set_points(X)
inv, mu, w, cov, lls, conv = ctx.flat_train(20, 0.0, mu0, cov0, w0, "diag", "W")
lr = ctx.empty((len(X), 800), np.float32)
ctx.flat_estep(inv, mu, w, out=lr)
dmu = ctx.to_device(mu)
def t(label, fn, reps=8):
    for _ in range(2): fn()
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.synchronize(); print("%-50s %.3f ms" % (label, (time.perf_counter() - t0) / reps * 1e3), flush=True)
t("mstep host out, host hint", lambda: ctx.flat_mstep(lr.exp(), centre_hint=mu))
t("mstep host out, no hint", lambda: ctx.flat_mstep(lr.exp()))
t("mstep device out, device hint", lambda: ctx.flat_mstep(lr.exp(), centre_hint=dmu, device_out=True))
t("mstep device out, no hint", lambda: ctx.flat_mstep(lr.exp(), device_out=True))
t("mstep host out, host hint (again)", lambda: ctx.flat_mstep(lr.exp(), centre_hint=mu))
