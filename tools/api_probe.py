#!/usr/bin/env python3
"""Wall time of every public call of the hot path at its benchmark size, beside the time its kernels take (hipEvents,
where the library times them): what a caller pays outside the kernels -- allocation, staging, synchronisation."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hgmm_amd
import bench
from oracle import hgmm_tree

ctx = hgmm_amd.Context(0)


def timeit(label, fn, reps=10, kernel=None):
    # (the first series of calls of a kind carries one-time costs -- code objects, buffers, queues: 8 ms spread over the
    #  first M-step series when it had two warm-up calls -- so every line gets four untimed calls first)
    # (... and they run with the kernel timers already on: the first timed M-step of a process pays a one-time 8 ms in
    #  the runtime when its event pairs are first read -- per call 0.52 - 0.55 ms with the timers off, on, and off again,
    #  except for that one call; profiles/r04/README.md)
    if kernel:
        ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(4):
        fn()
    ctx.synchronize()
    if kernel:
        ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    extra = ""
    if kernel:
        ctx.profile_enable(False)
        ms, n = ctx.profile_get(kernel)
        extra = "  kernels[%s] %.3f ms per call" % (kernel, ms / reps)
    print("%-64s %9.3f ms%s" % (label, dt * 1e3, extra), flush=True)


X = bench.synth_frame(0)
mu0, w0, cov0 = bench.init_params(X)
t0 = time.perf_counter(); ctx.set_points(X); ctx.synchronize()
print("%-64s %9.3f ms" % ("set_points float32 [1M,3] (first)", (time.perf_counter() - t0) * 1e3))
timeit("set_points float32 [1M,3]", lambda: ctx.set_points(X), 5)
X64 = X.astype(np.float64)
timeit("set_points float64 [1M,3]", lambda: ctx.set_points(X64), 5)
ctx.set_points(X)
inv, mu, w, cov, lls, conv = ctx.flat_train(20, 0.0, mu0, cov0, w0, "diag", "W")
timeit("flat_estep (allocating log_resp each call)", lambda: ctx.flat_estep(inv, mu, w), 6, "flat_estep")
lr = ctx.empty((len(X), 800), np.float32)
timeit("flat_estep out=lr", lambda: ctx.flat_estep(inv, mu, w, out=lr), 6, "flat_estep")
timeit("flat_estep want_lpn + want_argmax, out=lr", lambda: ctx.flat_estep(inv, mu, w, out=lr, want_lpn=True, want_argmax=True), 6, "flat_estep")
timeit("flat_estep no log_resp, want_lpn", lambda: ctx.flat_estep(inv, mu, w, want_log_resp=False, want_lpn=True), 6, "flat_estep")
timeit("flat_log_prob", lambda: ctx.flat_log_prob(inv, mu), 6, "flat_estep")
timeit("flat_predict", lambda: ctx.flat_predict(inv, mu, w), 10, "flat_estep")
timeit("flat_predict(...).get() (labels to the host)", lambda: ctx.flat_predict(inv, mu, w).get(), 10, "flat_estep")
timeit("flat_mstep(lr.exp()) host outputs", lambda: ctx.flat_mstep(lr.exp(), centre_hint=mu), 6, "flat_mstep")
timeit("flat_stats", lambda: ctx.flat_stats(inv, mu, w), 6, "flat_fused")
timeit("flat_train 20 iterations (upload, loop, download)", lambda: ctx.flat_train(20, 0.0, mu0, cov0, w0, "diag", "W"), 3, "flat_fused")
del lr
# ---- module-level functions on a resident cloud (DevicePoints): array module in = array module out
from hgmm_amd.gmm_waymo import gmm_impl as W
hgmm_amd.set_default_context(ctx)
dX = W.asarray(X)
d_inv, d_mu, d_w = ctx.to_device(inv), ctx.to_device(mu), ctx.to_device(w)
timeit("W.predict(DevicePoints, device params) -> DeviceArray labels", lambda: W.predict(dX, d_inv, d_mu, d_w), 10, "flat_estep")
timeit("W.predict(DevicePoints, host params)   -> DeviceArray labels", lambda: W.predict(dX, inv, mu, w), 10, "flat_estep")
timeit("W.predict(host X [1M,3], host params)  -> NumPy int64 (upload + download)", lambda: W.predict(X, inv, mu, w), 5, "flat_estep")
timeit("W.e_step(DevicePoints, device params) (lazy mean)", lambda: W.e_step(dX, d_inv, d_mu, d_w), 6, "flat_estep")
other = W.asarray(X[::2])
timeit("re-binding between two resident clouds + predict each", lambda: (W.predict(dX, d_inv, d_mu, d_w), W.predict(other, d_inv, d_mu, d_w)), 10, "flat_estep")
other.free(); dX.free()
ctx.set_points(X)
# ---- bunny-sized module-level API
B = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "bun000_xyz.npy"))
from hgmm_amd.gmm_waymo.gmm import GMM_GPU
g = GMM_GPU(n_gmm_components=100, max_iter=20, tol=0.0)
g.init()
timeit("GMM_GPU(100, max_iter=20).compute(bun000)", lambda: g.compute(B.astype(np.float32)), 5)
timeit("GMM_GPU.predict(bun000)", lambda: g.predict(B.astype(np.float32)), 5)
# ---- tree, call by call
P = B.astype(np.float64)
ctx.set_points(P)
L = 3; T = hgmm_tree.n_total(L); idx = np.random.RandomState(72).randint(T, size=T)
pi, mu_t, cov_t, leaf, iters, q = ctx.tree_build(L, 80.0, 1e-4, P[idx], 0.00034, 1000)
timeit("tree_build L=3 bun000", lambda: ctx.tree_build(L, 80.0, 1e-4, P[idx], 0.00034, 1000), 5)
par = np.full(len(P), -1, np.int32)
timeit("tree_estep (level 0, API-granular)", lambda: ctx.tree_estep(pi, mu_t, cov_t, par), 5)
timeit("tree_loglik level 2 (API-granular)", lambda: ctx.tree_loglik(pi, mu_t, cov_t, hgmm_tree.level(2), hgmm_tree.level(3)), 5)
ctx.tree_set_nodes(L, pi, mu_t, cov_t)
tgt = P[::2] + 0.001
timeit("tree_set_target (20k points)", lambda: ctx.tree_set_target(tgt), 5)
timeit("tree_reg_estep (moments to the host)", lambda: ctx.tree_reg_estep(T), 10, "tree_reg")
timeit("tree_reg_normal (28 numbers to the host)", lambda: ctx.tree_reg_normal(), 10, "tree_reg")
R0, t0v = np.identity(3), np.zeros(3)
timeit("tree_register 20 iterations", lambda: ctx.tree_register(R0.copy(), t0v.copy(), 1.0, 0.01, 20, 0.0), 5, "tree_reg")
timeit("fullcov_estep J=100 on bun000", lambda: ctx.fullcov_estep(np.full(100, 0.01), P[:100].copy(), np.tile(np.identity(3) * 1e-4, (100, 1, 1))), 5)
# ---- k-means step
ctx.set_points(X64 - X64.mean(0))
cen = (X64 - X64.mean(0))[np.random.RandomState(1).choice(len(X64), 800, replace=False)]
ctx.kmeans_step(cen, reset_labels=True)
timeit("kmeans_step k=800 on 1M points", lambda: ctx.kmeans_step(cen), 5, "kmeans_assign")
