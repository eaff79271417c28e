"""The N > 1 data path: two processes, each with its own context and its own shard of the cloud.

  * on ONE GPU the ranks are joined by the host shared-memory communicator (hgmm_comm_init_host -- the same
    all-reduce call sites as RCCL, which does not accept two ranks on one device);
  * with >= 2 GPUs visible the same comparison runs over RCCL (one rank per GPU, unique id shipped over the
    package's plain-TCP bootstrap) -- skipped on a single-GPU box.

Every sharded fit must reproduce the single-context fit on the whole cloud: same iteration counts and hard
labels, parameters equal to summation-order noise."""
import multiprocessing as mp
import os

import numpy as np
import pytest

from oracle import flat_em, hgmm_tree

pytestmark = pytest.mark.gpu

N_ALL, SPLIT = 6000, 2300          # uneven shards


def _cloud():
    rs = np.random.RandomState(17)
    centres = rs.rand(9, 3)
    return centres[rs.randint(9, size=N_ALL)] + 0.03 * rs.randn(N_ALL, 3)


def _inputs():
    X = _cloud()
    X32 = X.astype(np.float32)
    mu0, w0, cov0 = flat_em.seeded_init(X32, 70, 3)
    T = hgmm_tree.n_total(2)
    tree_init = X[np.random.RandomState(5).randint(N_ALL, size=T)]
    full_init = X[np.random.RandomState(6).choice(N_ALL, 8, replace=False)]
    km_init = X[np.random.RandomState(7).choice(N_ALL, 12, replace=False)]
    th = np.deg2rad(7.0)
    rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    target = X @ rz.T + 0.01
    return X, X32, (mu0, w0, cov0), tree_init, full_init, km_init, target


def _run_all(ctx, lo, hi):
    """Everything one rank (or the single context, lo = 0, hi = N) computes."""
    from hgmm_amd.kmeans import KMeans
    X, X32, (mu0, w0, cov0), tree_init, full_init, km_init, target = _inputs()
    out = {}
    ctx.set_points(X32[lo:hi])
    for variant in ("W", "G"):
        inv, mu, w, cov, lls, _ = ctx.flat_train(6, 0.0, mu0, cov0, w0, "diag", variant)
        out["flat_" + variant] = (mu, w, cov, np.asarray(lls))
    ctx.set_points(X[lo:hi])
    pi, mu, cov, leaf, iters, q = ctx.tree_build(2, 5.0, 1e-4, tree_init, 0.01, 60)
    out["tree"] = (pi, mu, cov, leaf, np.asarray(iters), np.asarray(q))
    ctx.tree_set_target(target[lo:hi])
    out["reg"] = ctx.tree_reg_estep(len(pi), lambda_c=0.01)
    pi, mu, cov, labels, q = ctx.fullcov_fit(8, 1.0, 1e-4, full_init, 0.01, 40)
    out["full"] = (pi, mu, cov, labels, np.asarray(q))
    km = KMeans(n_clusters=12, init=km_init, max_iter=50, ctx=ctx).fit(X[lo:hi])
    out["kmeans"] = (km.cluster_centers_, km.labels_, km.n_iter_, km.inertia_)
    return out


def _worker(rank, name, q, rccl_port=None):
    import hgmm_amd
    if rccl_port is None:
        ctx = hgmm_amd.Context(0)
        ctx.comm_init_host(2, rank, name)
    else:
        from hgmm_amd import parallel
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(rccl_port), RANK=str(rank), LOCAL_RANK=str(rank),
                          WORLD_SIZE="2")
        ctx = hgmm_amd.Context(rank)                       # one rank per GPU
        parallel.attach_communicator(ctx, rank, 2, transport="tcp")
    lo, hi = (0, SPLIT) if rank == 0 else (SPLIT, N_ALL)
    res = _run_all(ctx, lo, hi)
    total = ctx.allreduce([float(hi - lo)])[0]
    ctx.close()
    q.put((rank, res, total))


def test_two_ranks_on_one_gpu_match_the_single_context_fit():
    _two_ranks_match_single_context(None)


def test_rccl_two_ranks_two_gpus():
    """Same comparison over RCCL / xGMI when the box has two GPUs (the driver's 8-GPU node; a 1-GPU box skips)."""
    import ctypes
    import socket
    import hgmm_amd
    cnt = ctypes.c_int(0)
    hgmm_amd.load_library().hgmm_device_count(ctypes.byref(cnt))
    if cnt.value < 2:
        pytest.skip("needs >= 2 GPUs (found %d)" % cnt.value)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    _two_ranks_match_single_context(port)


def _two_ranks_match_single_context(rccl_port):
    import hgmm_amd
    name = "hgmm_test_%d" % os.getpid()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_worker, args=(r, name, q, rccl_port)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in procs:
        rank, res, total = q.get(timeout=300)
        got[rank] = res
        assert total == N_ALL
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ctx = hgmm_amd.Context(0)
    ref = _run_all(ctx, 0, N_ALL)
    ctx.close()
    cut = {0: slice(0, SPLIT), 1: slice(SPLIT, N_ALL)}
    for rank in (0, 1):
        r = got[rank]
        for variant in ("W", "G"):
            mu, w, cov, lls = r["flat_" + variant]
            rmu, rw, rcov, rlls = ref["flat_" + variant]
            np.testing.assert_allclose(lls, rlls, rtol=0, atol=2e-6)
            np.testing.assert_allclose(mu, rmu, rtol=0, atol=2e-6)
            np.testing.assert_allclose(w, rw, rtol=1e-5, atol=1e-8)
            np.testing.assert_allclose(cov, rcov, rtol=1e-4, atol=1e-9)
        pi, mu, cov, leaf, iters, qt = r["tree"]
        rpi, rmu, rcov, rleaf, riters, rq = ref["tree"]
        assert list(iters) == list(riters) and np.array_equal(leaf, rleaf[cut[rank]])
        np.testing.assert_allclose(qt, rq, rtol=1e-11)
        np.testing.assert_allclose(pi, rpi, rtol=0, atol=1e-12)
        np.testing.assert_allclose(mu, rmu, rtol=0, atol=1e-11)
        np.testing.assert_allclose(cov, rcov, rtol=1e-8, atol=1e-14)
        for a, b in zip(r["reg"], ref["reg"]):
            np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-12)
        pi, mu, cov, labels, qt = r["full"]
        rpi, rmu, rcov, rlabels, rq = ref["full"]
        assert len(qt) == len(rq) and np.array_equal(labels, rlabels[cut[rank]])
        np.testing.assert_allclose(qt, rq, rtol=1e-11)
        np.testing.assert_allclose(mu, rmu, rtol=0, atol=1e-10)
        centres, labels, n_iter, inertia = r["kmeans"]
        rc, rl, rn, ri = ref["kmeans"]
        assert n_iter == rn and np.array_equal(labels, rl[cut[rank]])
        np.testing.assert_allclose(centres, rc, rtol=0, atol=1e-12)
        np.testing.assert_allclose(inertia, ri, rtol=1e-10)
    # both ranks hold identical models
    for key in ("flat_W", "tree", "full"):
        for a, b in zip(got[0][key][:3], got[1][key][:3]):
            assert np.array_equal(a, b)
