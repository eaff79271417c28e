"""Several contexts alive at once in ONE process: the replica fan-out of independent scan pairs (north_star:
"independent scan pairs ... fan out across the GPUs", the reference's unit of work is one pair,
src/python/hgmm/hgmm_gpu.py:802-807, gmmreg_gpu/gmmreg.py:149-157) drives one context per GPU from one launcher, and
include/hgmm.h promises "distinct contexts may be driven from distinct threads".

  * two contexts on one GPU driven from two threads concurrently (ctypes drops the GIL inside every library call):
    flat fit, tree build, full-covariance fit, KMeans and a registration each -- results equal the serial ones;
  * one thread interleaving the split training loop (begin / step / end) of two contexts;
  * the same two shapes with the contexts on two DIFFERENT devices (skipped on a one-GPU box): every C-ABI entry selects
    its context's device itself (HGMM_ENTER, csrc/hgmm_ctx.h), so the calling thread's current device never matters.
"""
import ctypes
import threading

import numpy as np
import pytest

from oracle import flat_em, hgmm_tree

pytestmark = pytest.mark.gpu

N = 20000


def _gpu_count():
    import hgmm_amd
    cnt = ctypes.c_int(0)
    hgmm_amd.load_library().hgmm_device_count(ctypes.byref(cnt))
    return cnt.value


def _job(seed):
    rs = np.random.RandomState(seed)
    centres = rs.rand(11, 3)
    X = centres[rs.randint(11, size=N)] + 0.03 * rs.randn(N, 3)
    X32 = X.astype(np.float32)
    mu0, w0, cov0 = flat_em.seeded_init(X32, 64, seed)
    T = hgmm_tree.n_total(2)
    tree_init = X[np.random.RandomState(seed + 1).randint(N, size=T)]
    full_init = X[np.random.RandomState(seed + 2).choice(N, 8, replace=False)]
    km_init = X[np.random.RandomState(seed + 3).choice(N, 12, replace=False)]
    th = np.deg2rad(5.0 + seed)
    rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    target = X[::3] @ rz.T + 0.01
    return X, X32, (mu0, w0, cov0), tree_init, full_init, km_init, target


def _run(ctx, job, reps=2):
    """Everything one replica computes, `reps` times over (the second pass reuses every buffer of the first)."""
    from hgmm_amd.kmeans import KMeans
    X, X32, (mu0, w0, cov0), tree_init, full_init, km_init, target = job
    out = None
    for _ in range(reps):
        out = {}
        ctx.set_points(X32)
        inv, mu, w, cov, lls, _ = ctx.flat_train(8, 0.0, mu0, cov0, w0, "diag", "W")
        out["flat"] = (mu, w, cov, np.asarray(lls))
        out["labels"] = ctx.flat_predict(inv, mu, w, "diag", "W").get()
        ctx.set_points(X)
        pi, mu, cov, leaf, iters, q = ctx.tree_build(2, 5.0, 1e-4, tree_init, 0.01, 60)
        out["tree"] = (pi, mu, cov, leaf, np.asarray(iters), np.asarray(q))
        ctx.tree_set_target(target)
        rot, t, n_it, q_reg, status, _ = ctx.tree_register(np.eye(3), np.zeros(3), 1.0, 0.01, 10, 1e-9)
        out["reg"] = (rot, t, np.float64(np.nan if q_reg is None else q_reg), np.int64(n_it), np.int64(status))
        pi, mu, cov, labels, q = ctx.fullcov_fit(8, 1.0, 1e-4, full_init, 0.01, 40)
        out["full"] = (pi, mu, cov, labels, np.asarray(q))
        km = KMeans(n_clusters=12, init=km_init, max_iter=50, ctx=ctx).fit(X)
        out["kmeans"] = (km.cluster_centers_, km.labels_, np.int64(km.n_iter_), np.float64(km.inertia_))
    return out


def _same(a, b, what):
    for key in a:
        for i, (x, y) in enumerate(zip(a[key], b[key])):
            x, y = np.asarray(x), np.asarray(y)
            assert x.shape == y.shape, (what, key, i)
            if x.dtype.kind in "iu":
                assert np.array_equal(x, y), (what, key, i)
            else:
                # serial runs of one context are reproducible to the last bit in every family but the ones whose
                # statistics are summed with floating-point atomics: equal to summation-order noise
                np.testing.assert_allclose(x, y, rtol=1e-9, atol=1e-11 if x.dtype == np.float64 else 2e-6,
                                           err_msg="%s %s[%d]" % (what, key, i))


def _threads(ctxs, jobs):
    res, errs = [None] * len(ctxs), []
    gate = threading.Barrier(len(ctxs))

    def work(i):
        try:
            gate.wait(30)
            res[i] = _run(ctxs[i], jobs[i])
        except BaseException as e:                             # noqa: BLE001 -- reported by the asserting thread
            errs.append((i, repr(e)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(ctxs))]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    assert not errs, errs
    assert all(r is not None for r in res)
    return res


def _serial(device, jobs):
    import hgmm_amd
    out = []
    for job in jobs:
        ctx = hgmm_amd.Context(device)
        out.append(_run(ctx, job, reps=1))
        ctx.close()
    return out


def test_two_contexts_one_gpu_two_threads_match_the_serial_runs():
    import hgmm_amd
    jobs = [_job(1), _job(2)]
    ref = _serial(0, jobs)
    ctxs = [hgmm_amd.Context(0), hgmm_amd.Context(0)]
    got = _threads(ctxs, jobs)
    for c in ctxs:
        c.close()
    for i in range(2):
        _same(got[i], ref[i], "thread %d" % i)


def test_four_contexts_one_gpu_four_threads():
    """More replicas than a pair: four contexts, four threads, four different clouds."""
    import hgmm_amd
    jobs = [_job(s) for s in (3, 4, 5, 6)]
    ref = _serial(0, jobs)
    ctxs = [hgmm_amd.Context(0) for _ in jobs]
    got = _threads(ctxs, jobs)
    for c in ctxs:
        c.close()
    for i in range(len(jobs)):
        _same(got[i], ref[i], "thread %d" % i)


def _interleaved_split_loops(devices):
    """ONE thread, two contexts, the split loop's calls alternating between them (flat_train_step does not synchronise:
    both streams hold work at the same time)."""
    import hgmm_amd
    jobs = [_job(7), _job(8)]
    ref = []
    for dev, job in zip(devices, jobs):
        ctx = hgmm_amd.Context(dev)
        ctx.set_points(job[1])
        mu0, w0, cov0 = job[2]
        ref.append(ctx.flat_train(12, 0.0, mu0, cov0, w0, "diag", "W"))
        ctx.close()
    ctxs = [hgmm_amd.Context(d) for d in devices]
    for ctx, job in zip(ctxs, jobs):
        ctx.set_points(job[1])
    for ctx, job in zip(ctxs, jobs):
        mu0, w0, cov0 = job[2]
        ctx.flat_train_begin(0.0, mu0, cov0, w0, "diag", "W", lls_capacity=64)
    for _ in range(4):
        for ctx in ctxs:
            ctx.flat_train_step(3)
    for ctx, r in zip(ctxs, ref):
        inv, mu, w, cov, lls, conv, n_it = ctx.flat_train_end()
        assert n_it == 12
        for x, y in zip((inv, mu, w, cov, np.asarray(lls)), (r[0], r[1], r[2], r[3], np.asarray(r[4]))):
            np.testing.assert_allclose(x, y, rtol=1e-5, atol=2e-6)
    # device buffers of one context handed to calls of that context while the OTHER context's device was used last
    lr = [ctx.flat_estep(r[0], r[1], r[2], "diag", "W")[1] for ctx, r in zip(ctxs, ref)]
    for ctx, a, job, r in zip(ctxs, lr, jobs, ref):
        host = a.get_rows(0, 64)
        _, o_lr = flat_em.e_step(job[1][:64].astype(np.float64), r[0].astype(np.float64), r[1].astype(np.float64),
                                 r[2].astype(np.float64), "diag", "W")
        assert np.abs(np.exp(host.astype(np.float64)) - np.exp(o_lr)).max() < 1e-5
    for a in lr:
        a.free()
    for ctx in ctxs:
        ctx.close()


def test_one_thread_interleaving_two_contexts_on_one_gpu():
    _interleaved_split_loops([0, 0])


def test_one_thread_interleaving_contexts_on_two_gpus():
    if _gpu_count() < 2:
        pytest.skip("needs >= 2 GPUs (found %d)" % _gpu_count())
    _interleaved_split_loops([0, 1])


def test_contexts_on_all_gpus_from_threads():
    """One context per GPU on min(8, visible GPUs) devices, one thread each -- the replica launcher's shape."""
    import hgmm_amd
    n = _gpu_count()
    if n < 2:
        pytest.skip("needs >= 2 GPUs (found %d)" % n)
    n = min(8, n)
    jobs = [_job(10 + i) for i in range(n)]
    ref = _serial(0, jobs)
    ctxs = [hgmm_amd.Context(i) for i in range(n)]
    got = _threads(ctxs, jobs)
    for c in ctxs:
        c.close()
    for i in range(n):
        _same(got[i], ref[i], "device %d" % i)


# ---- the replica pool: independent scan pairs / frames fanned out over contexts (hgmm_amd.replicas) ----------------
def _scan_pairs(bunny, count):
    a = bunny.astype(np.float64)[::4]
    out = []
    for k in range(count):
        rs = np.random.RandomState(50 + k)
        axis = rs.randn(3)
        axis /= np.linalg.norm(axis)
        th = np.deg2rad(rs.uniform(3.0, 9.0))
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
        out.append((a, a[::-1] @ R.T + rs.uniform(-0.005, 0.005, 3)))
    return out


def test_register_pairs_over_a_replica_pool_matches_the_serial_calls(bunny):
    """register_pairs = registration_gmmtree per pair (src/python/hgmm/hgmm_gpu.py:802-807), the pairs handed out to
    several contexts (here: three on one GPU; on a multi-GPU node also one per device): same transformation per pair
    as the serial calls, in the order of the input."""
    import hgmm_amd
    from hgmm_amd.hgmm.hgmm_gpu import registration_gmmtree
    from hgmm_amd.replicas import register_pairs, device_count
    pairs = _scan_pairs(bunny, 7)
    kw = dict(tree_level=2, lambda_c=0.01, ls=20, sig2=0.004)
    ctx = hgmm_amd.Context(0)
    serial = [registration_gmmtree(s, t, maxiter=15, tol=1e-6, ctx=ctx, **kw) for s, t in pairs]
    ctx.close()
    for devices, per in (([0], 3), (None, 1)):
        got = register_pairs(pairs, devices=devices, contexts_per_device=per, maxiter=15, tol=1e-6, **kw)
        assert len(got) == len(pairs)
        for g, r, (s, t) in zip(got, serial, pairs):
            np.testing.assert_allclose(g.transformation.rot, r.transformation.rot, rtol=0, atol=1e-9)
            np.testing.assert_allclose(g.transformation.t, r.transformation.t, rtol=0, atol=1e-10)
            # and it is a registration: the source lands on the target (a permuted, moved copy of itself)
            err = np.linalg.norm(g.transformation.transform(s) - t[::-1], axis=1).mean()
            assert err < 2e-3, err
    assert device_count() >= 1


def test_fit_frames_over_a_replica_pool_and_module_level_api_in_threads(bunny):
    """fit_frames = GMM_GPU_Base.fit per frame (gmm_waymo/src/gmm.py:65-84) on the worker's own context: the
    module-level functions of the mirrors (train_gmm, init_gmm_params, asarray) pick the THREAD's context up through
    use_context, so nothing runs on the process-wide default context."""
    import hgmm_amd
    from hgmm_amd.gmm_waymo import gmm_impl
    from hgmm_amd.replicas import fit_frames, ReplicaPool
    frames = [bunny[k::5].copy() for k in range(5)]
    np.random.seed(4)
    inits = [gmm_impl.init_gmm_params(f, 40, "diag") for f in frames]
    ctx = hgmm_amd.Context(0)
    ref = []
    for f, (mu, w, cov) in zip(frames, inits):
        ctx.set_points(f)
        ref.append(ctx.flat_train(10, 0.0, mu, cov, w, "diag", "W"))
    ctx.close()
    with ReplicaPool(devices=[0], contexts_per_device=3) as pool:
        def one(c, job):
            f, (mu, w, cov) = job
            assert hgmm_amd.default_context() is c
            return gmm_impl.train_gmm(f, 10, 0.0, mu, cov, w, cov_type="diag")   # host array in: uploaded to THIS thread's context
        got = pool.map(one, list(zip(frames, inits)))
        for g, r in zip(got, ref):
            np.testing.assert_allclose(g[1], r[1], rtol=0, atol=2e-6)           # means
            np.testing.assert_allclose(np.asarray(g[4]), np.asarray(r[4]), rtol=0, atol=2e-6)
        models = fit_frames(frames, n_components=40, max_iter=5, pool=pool)
        assert len(models) == 5 and all(m.means_.shape == (40, 3) and np.isfinite(m.lls).all() for m in models)


def test_register_pairs_gmmreg_over_a_replica_pool():
    """method="gmmreg": registration_gmmreg (gmmreg_gpu/gmmreg.py:149-157) per pair -- the mixtures' KMeans initialiser
    and EM fit, the Gauss transforms behind every BFGS cost evaluation: all of it on the WORKER's context (thread-local
    default context), several pairs in flight; the same transformations as the serial calls."""
    from conftest import load_golden
    from hgmm_amd.gmmreg_gpu.gmmreg import registration_gmmreg
    from hgmm_amd.replicas import register_pairs
    g = load_golden("gmmreg_l2.npz")
    src, tgt = g["reg_source"], g["reg_target"]
    th = np.deg2rad(4.0)
    rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    pairs = [(src, tgt), (src, tgt @ rz.T + 0.01), (src @ rz.T, tgt), (src, tgt)]
    serial = [registration_gmmreg(s, t, n_gmm_components=30) for s, t in pairs]
    got = register_pairs(pairs, devices=[0], contexts_per_device=2, method="gmmreg", n_gmm_components=30)
    assert len(got) == len(pairs)
    for a, b in zip(got, serial):
        np.testing.assert_allclose(a.rot, b.rot, rtol=0, atol=1e-7)
        np.testing.assert_allclose(a.t, b.t, rtol=0, atol=1e-7)
    np.testing.assert_allclose(got[0].rot, g["reg_rot"], atol=5e-3)
