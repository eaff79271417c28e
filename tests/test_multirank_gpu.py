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


def _bounds(world):
    """Cut points of `world` uneven contiguous shards of the N_ALL points (world = 2: the historical 2300 / 3700)."""
    if world == 2:
        return [0, SPLIT, N_ALL]
    return [0] + [int(N_ALL * ((r + 1) / world) ** 1.15) for r in range(world - 1)] + [N_ALL]


def _gpu_count():
    import ctypes
    import hgmm_amd
    cnt = ctypes.c_int(0)
    hgmm_amd.load_library().hgmm_device_count(ctypes.byref(cnt))
    return cnt.value


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


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
    before = ctx.comm_stats()
    pi, mu, cov, leaf, iters, q = ctx.tree_build(2, 5.0, 1e-4, tree_init, 0.01, 60)
    after = ctx.comm_stats()
    out["tree"] = (pi, mu, cov, leaf, np.asarray(iters), np.asarray(q))
    out["tree_collectives"] = (after[0] - before[0], after[1] - before[1])
    ctx.tree_set_target(target[lo:hi])
    out["reg"] = ctx.tree_reg_estep(len(pi), lambda_c=0.01)
    pi, mu, cov, labels, q = ctx.fullcov_fit(8, 1.0, 1e-4, full_init, 0.01, 40)
    out["full"] = (pi, mu, cov, labels, np.asarray(q))
    km = KMeans(n_clusters=12, init=km_init, max_iter=50, ctx=ctx).fit(X[lo:hi])
    out["kmeans"] = (km.cluster_centers_, km.labels_, km.n_iter_, km.inertia_)
    return out


def _worker(rank, name, q, rccl_port=None, backend="host", world=2):
    try:
        _worker_body(rank, name, q, rccl_port, backend, world)
    except BaseException as e:                              # the parent must not wait out its timeout for a dead rank
        import traceback
        q.put((rank, "rank %d failed: %r\n%s" % (rank, e, traceback.format_exc()), None))
        raise


def _worker_body(rank, name, q, rccl_port, backend, world):
    import hgmm_amd
    if rccl_port is None:
        ctx = hgmm_amd.Context(0)
        if backend == "ipc":
            ctx.comm_init_ipc(world, rank, name)
        else:
            ctx.comm_init_host(world, rank, name)
    else:
        from hgmm_amd import parallel
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(rccl_port), RANK=str(rank), LOCAL_RANK=str(rank),
                          WORLD_SIZE=str(world))
        ctx = hgmm_amd.Context(rank)                       # one rank per GPU
        parallel.attach_communicator(ctx, rank, world, transport="tcp")
    lo, hi = _bounds(world)[rank], _bounds(world)[rank + 1]
    res = _run_all(ctx, lo, hi)
    total = ctx.allreduce([float(hi - lo)])[0]
    ctx.close()
    q.put((rank, res, total))


def test_two_ranks_on_one_gpu_match_the_single_context_fit():
    _two_ranks_match_single_context(None)


def test_two_ranks_on_one_gpu_peer_exchange_backend():
    """The one-shot peer exchange (hgmm_comm_init_ipc: every rank writes its slice into every peer's mapped buffer, one
    kernel per all-reduce) behind the same call sites: two processes that share the box's GPU map each other's
    exchange buffers through hipIpc handles.  Same fits, same iteration counts, bitwise equal models on both ranks."""
    _two_ranks_match_single_context(None, backend="ipc")


def test_three_uneven_ranks_on_one_gpu_peer_exchange_backend():
    """world = 3 (uneven shards 1697 / 2066 / 2237) through the same code the multi-GPU tests below use with
    world = min(8, GPUs): a 1-GPU box thereby runs the world > 2 comparison logic too."""
    _ranks_match_single_context(None, backend="ipc", world=3)


def test_peer_exchange_collectives_two_ranks_one_gpu():
    """The exchange kernel by itself: sum / max of float64 and payloads larger than one slot (pieces), many collectives
    in a row (slot parity, flag sequence), both ranks bitwise equal and equal to the host's sum in rank order."""
    name = "hgmm_ipc_%d" % os.getpid()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_ipc_collectives_worker, args=(r, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in procs)
    assert not any(isinstance(v, str) for v in got.values()), got
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for a, b in zip(got[0], got[1]):
        assert np.array_equal(a, b)
    exp = _ipc_collectives_expected()
    assert len(exp) == len(got[0])
    for a, e in zip(got[0], exp):
        assert np.array_equal(a, e)


def test_peer_exchange_missing_peer_raises_instead_of_hanging():
    """A peer that never joins a collective must not hang the GPU: the exchange kernel gives up after its timeout
    (HGMM_IPC_TIMEOUT_S, 20 s by default; 2 s here), raises the context's error word, and the call that waits for the
    result fails; collectives already enqueued behind it drain at once."""
    import time
    name = "hgmm_ipc_to_%d" % os.getpid()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_ipc_timeout_worker, args=(r, name, q)) for r in range(2)]
    t0 = time.time()
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert got[1] == "idle"
    assert got[0].startswith("raised:") and ("peer exchange" in got[0] or "hipErrorLaunchTimeOut" in got[0] or "timed out" in got[0].lower()), got[0]
    assert time.time() - t0 < 100


def _ipc_timeout_worker(rank, name, q):
    os.environ["HGMM_IPC_TIMEOUT_S"] = "2"
    import time
    import hgmm_amd
    ctx = hgmm_amd.Context(0)
    ctx.comm_init_ipc(2, rank, name)
    if rank == 1:
        time.sleep(12)                                     # never takes part in the collective
        q.put((rank, "idle"))
        q.close(); q.join_thread()                         # (the queue's feeder thread must have sent it before _exit)
        os._exit(0)                                        # (no orderly teardown with a peer that has given up)
    try:
        t0 = time.time()
        ctx.allreduce(np.arange(5000.0))
        ctx.allreduce(np.arange(5000.0))                   # behind a failed collective: must not sit out another timeout
        q.put((rank, "no error after %.1f s" % (time.time() - t0)))
    except hgmm_amd.HgmmError as e:
        q.put((rank, "raised: %s (%.1f s)" % (e, time.time() - t0)))
    q.close(); q.join_thread()
    os._exit(0)


def test_configs4_eight_frames_joint_fit_matches_oracle_fixture():
    """BASELINE configs[4] ITSELF -- 8 frames x 1M points, J = 800, one frame per rank, one joint fit with an all-reduce of
    the sufficient statistics per iteration -- with the 8 ranks sharing the box's ONE GPU (joined by the one-shot peer
    exchange: each rank maps the seven other exchange buffers): the data path of the 8-GPU run at full size against
    oracle.flat_em's float64 EM on the 8M points together (tests/golden/flat_uniform8x1M_J800_oracle.npz,
    tools/gen_oracle_fixtures.py --only flat8x1m).  All eight ranks must end with bitwise the same model."""
    _full_frames_joint_fit("ipc", 8)


@pytest.mark.parametrize("backend", ["ipc", "host"])
def test_two_full_frames_joint_fit_matches_oracle_fixture(backend):
    """BASELINE configs[4] in small, at the bench's frame size: two ranks with one 1M-point frame each (bench.synth_frame(0)
    and (1), initial parameters from frame 0 as bench.py takes them) fit ONE mixture, all-reducing the 57 KB of
    sufficient statistics per iteration -- against oracle.flat_em's float64 EM on the 2M points together
    (tests/golden/flat_uniform2x1M_J800_oracle.npz, tools/gen_oracle_fixtures.py --only flat2x1m).  Both ranks must
    hold the same model (bitwise under the peer exchange, whose sums are rank-ordered on every rank)."""
    _full_frames_joint_fit(backend, 2)


def _full_frames_joint_fit(backend, world):
    port = _free_port() if backend == "rccl" else None
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "flat_uniform%dx1M_J800_oracle.npz" % world))
    assert int(g["frames"]) == world
    name = "hgmm_%dx1m_%s_%d" % (world, backend, os.getpid())
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_joint_fit_worker, args=(r, name, q, backend, world, port)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in procs)
    assert not any(isinstance(v, str) for v in got.values()), got
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank in range(world):
        mu, w, cov, inv, lls = got[rank]
        d_ll = np.abs(np.asarray(lls, dtype=np.float64) - g["lls"]).max()
        d_mu = np.abs(mu - g["mu"]).max()
        print("rank %d (%s): max|dlls| %.3g max|dmu| %.3g max rel dw %.3g max rel dcov %.3g"
              % (rank, backend, d_ll, d_mu, np.abs(w / g["w"] - 1).max(), np.abs(cov / g["cov"] - 1).max()))
        assert len(lls) == int(g["iters"]) and d_ll <= 2e-5 and d_mu <= 5e-6
        np.testing.assert_allclose(w, g["w"], rtol=1e-5, atol=1e-10)
        np.testing.assert_allclose(cov, g["cov"], rtol=1e-4, atol=1e-9)
        np.testing.assert_allclose(inv, g["inv"], rtol=1e-4)
    for rank in range(1, world):                              # (RCCL's all-reduce also leaves every rank with the same sums)
        for a, b in zip(got[0], got[rank]):
            assert np.array_equal(a, b)


def _joint_fit_worker(rank, name, q, backend, world=2, rccl_port=None):
    try:
        import hgmm_amd
        N, J = 1_000_000, 800
        frame0 = np.random.RandomState(0).rand(N, 3).astype(np.float32)
        frame = frame0 if rank == 0 else np.random.RandomState(rank).rand(N, 3).astype(np.float32)
        idx = np.random.RandomState(100).choice(N, J, replace=False)
        mu0 = frame0[idx].copy()
        w0 = (np.ones(J) / J).astype(np.float32)
        cov0 = (0.1 * np.ones((J, 3))).astype(np.float32)
        if backend == "rccl":                                 # one rank per GPU, RCCL over xGMI
            from hgmm_amd import parallel
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(rccl_port), RANK=str(rank), LOCAL_RANK=str(rank),
                              WORLD_SIZE=str(world))
            ctx = hgmm_amd.Context(rank)
            parallel.attach_communicator(ctx, rank, world, transport="tcp")
        else:
            ctx = hgmm_amd.Context(0)
            (ctx.comm_init_ipc if backend == "ipc" else ctx.comm_init_host)(world, rank, name)
        ctx.set_points(frame)
        inv, mu, w, cov, lls, _ = ctx.flat_train(3, 0.0, mu0, cov0, w0, "diag", "W")
        ctx.comm_destroy()
        ctx.close()
        q.put((rank, (mu, w, cov, inv, np.asarray(lls))))
    except BaseException as e:
        q.put((rank, "rank %d failed: %r" % (rank, e)))
        raise


def _ipc_payloads(rank):
    rs = np.random.RandomState(40 + rank)
    return [rs.randn(n) * 10.0 ** rs.randint(-3, 4) for n in (1, 7, 511, 512, 513, 7170, 65536, 65537, 150001)]


def _ipc_collectives_expected(world=2):
    """What the exchange must return: the slices added up in RANK ORDER (((r0 + r1) + r2) + ...), maxima likewise."""
    per_rank = [_ipc_payloads(r) for r in range(world)]
    out = []
    for rep in range(3):
        for i in range(len(per_rank[0])):
            tot = per_rank[0][i] + rep
            mx = per_rank[0][i] + rep
            for r in range(1, world):
                tot = tot + (per_rank[r][i] + rep)
                mx = np.maximum(mx, per_rank[r][i] + rep)
            out.append(tot)
            out.append(mx)
    return out


def _ipc_collectives_worker(rank, name, q, device=0, world=2):
    try:
        import hgmm_amd
        ctx = hgmm_amd.Context(device)
        ctx.comm_init_ipc(world, rank, name)
        out = []
        for rep in range(3):
            for x in _ipc_payloads(rank):
                out.append(ctx.allreduce(x + rep))
                out.append(ctx.allreduce(x + rep, op="max"))
        ctx.comm_destroy()
        ctx.close()
        q.put((rank, out))
    except BaseException as e:
        q.put((rank, "rank %d failed: %r" % (rank, e)))
        raise


def test_rccl_ranks_on_all_gpus():
    """Same comparison over RCCL / xGMI, one rank per GPU on min(8, visible GPUs) of them (the driver's 8-GPU node runs
    world = 8; a 1-GPU box skips): every family's sharded fit against the single-context fit on the whole cloud."""
    n = _gpu_count()
    if n < 2:
        pytest.skip("needs >= 2 GPUs (found %d)" % n)
    _ranks_match_single_context(_free_port(), backend="rccl", world=min(8, n))


def test_rccl_two_ranks_two_gpus():
    """The two-rank case by itself (what a 2-GPU box can run; the test above covers it on larger nodes with world > 2)."""
    n = _gpu_count()
    if n < 2:
        pytest.skip("needs >= 2 GPUs (found %d)" % n)
    _ranks_match_single_context(_free_port(), backend="rccl", world=2)


def test_peer_exchange_ranks_on_all_gpus():
    """The one-shot peer exchange ACROSS min(8, visible GPUs) GPUs (each rank maps every other rank's exchange buffer
    over xGMI through its hipIpc handle): the collectives by themselves, bitwise equal on all ranks and equal to the
    host's sums in rank order.  Needs two GPUs (the driver's multi-GPU node; a 1-GPU box skips)."""
    n = _gpu_count()
    if n < 2:
        pytest.skip("needs >= 2 GPUs (found %d)" % n)
    world = min(8, n)
    name = "hgmm_ipc%d_%d" % (world, os.getpid())
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_ipc_collectives_worker, args=(r, name, q, r, world)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in procs)
    setup = [v for v in got.values() if isinstance(v, str) and "peer exchange:" in v and ("hipIpc" in v or "uncached" in v or "map" in v)]
    if setup:
        for p in procs:
            p.join(60)
        # the backend is optional (bench.py falls back, collectively): a node whose driver cannot map peer memory this
        # way is reported, not failed -- WRONG SUMS below are failures
        pytest.xfail("peer buffers cannot be mapped across GPUs on this node: %s" % setup[0])
    assert not any(isinstance(v, str) for v in got.values()), got
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    exp = _ipc_collectives_expected(world)
    for rank in range(world):
        assert len(got[rank]) == len(exp)
        for a, e in zip(got[rank], exp):
            assert np.array_equal(a, e)


def test_configs4_over_rccl_on_eight_gpus():
    """BASELINE configs[4] AS WRITTEN: 8 frames x 1M points, J = 800, one frame per GPU on 8 x MI355X, the (7 J + 2)
    float64 sufficient statistics all-reduced by RCCL over xGMI before every M-step -- against oracle.flat_em's float64
    EM on the 8M points together (tests/golden/flat_uniform8x1M_J800_oracle.npz).  Skips below 8 GPUs (the 8-ranks-on-
    one-GPU test above runs the same data path through the peer exchange there)."""
    n = _gpu_count()
    if n < 8:
        pytest.skip("needs 8 GPUs (found %d)" % n)
    _full_frames_joint_fit("rccl", 8)


def test_two_full_frames_over_rccl_on_two_gpus():
    """configs[4] in small over RCCL: two 1M-point frames on two GPUs against the 2M-point oracle fixture."""
    n = _gpu_count()
    if n < 2:
        pytest.skip("needs >= 2 GPUs (found %d)" % n)
    _full_frames_joint_fit("rccl", 2)


def _two_ranks_match_single_context(rccl_port, backend="host"):
    _ranks_match_single_context(rccl_port, backend, 2)


def _ranks_match_single_context(rccl_port, backend="host", world=2):
    import hgmm_amd
    name = "hgmm_test_%s_%d" % (backend, os.getpid())
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_worker, args=(r, name, q, rccl_port, backend, world)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in procs:
        rank, res, total = q.get(timeout=300)
        assert total is not None, res
        got[rank] = res
        assert total == N_ALL
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ctx = hgmm_amd.Context(0)
    ref = _run_all(ctx, 0, N_ALL)
    ctx.close()
    b = _bounds(world)
    cut = {r: slice(b[r], b[r + 1]) for r in range(world)}
    for rank in range(world):
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
        # No blind batches behind a level's stop: every rank leaves a level with exactly min(it + 2, budget)
        # iterations enqueued -- two all-reduces each (moments, q) -- plus the one all-reduce of the point count; the same
        # number on every rank (a mismatch would hang RCCL), and the single context enqueues none.
        enq = [min(int(it) + 2, 60) for it in iters]
        assert r["tree_collectives"] == (1 + 2 * sum(enq), sum(enq) - int(sum(iters))), (r["tree_collectives"], list(iters))
        assert ref["tree_collectives"] == (0, 0)
        for a, b_ in zip(r["reg"], ref["reg"]):
            np.testing.assert_allclose(a, b_, rtol=1e-10, atol=1e-12)
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
    # all ranks hold identical models
    for key in ("flat_W", "flat_G", "tree", "full") if backend == "ipc" else ("flat_W", "tree", "full"):
        for rank in range(1, world):
            for a, b_ in zip(got[0][key][:3], got[rank][key][:3]):
                assert np.array_equal(a, b_)


# ---- a stale shared-memory object of a crashed run must not be joined ---------------------------------------
def _stale_name_worker(rank, name, q, backend, delay):
    try:
        import time
        import hgmm_amd
        time.sleep(delay)
        ctx = hgmm_amd.Context(0)
        (ctx.comm_init_ipc if backend == "ipc" else ctx.comm_init_host)(2, rank, name)
        out = ctx.allreduce(np.arange(1000.0) * (rank + 1))
        ctx.comm_destroy()
        ctx.close()
        q.put((rank, out))
    except BaseException as e:
        q.put((rank, "rank %d failed: %r" % (rank, e)))
        raise


@pytest.mark.parametrize("backend", ["host", "ipc"])
def test_an_orphaned_shared_memory_object_is_not_joined(backend):
    """A run that crashed left /dev/shm/<name> behind, marked ready.  Rank 1 of the next run with the same name arrives
    BEFORE rank 0 has replaced the object: it must recognise the orphan (its creator's pid is dead), wait for rank 0's
    fresh object and join that one -- not publish into the orphan and sit out the barrier's timeout."""
    import struct
    import subprocess
    import sys
    import time
    name = "hgmm_stale_%s_%d" % (backend, os.getpid())
    dead = subprocess.Popen([sys.executable, "-c", "pass"])
    dead.wait()
    pidns = os.stat("/proc/self/ns/pid").st_ino
    path = "/dev/shm/" + name
    with open(path, "wb") as f:                               # {ready = 1, owner_pid = a dead one, its namespace, created now}
        f.write(struct.pack("<iiQQ", 1, dead.pid, pidns, int(time.time())))
        f.truncate(64 << 20)
    try:
        mpc = mp.get_context("spawn")
        q = mpc.Queue()
        procs = [mpc.Process(target=_stale_name_worker, args=(r, name, q, backend, 1.5 if r == 0 else 0.0)) for r in range(2)]
        t0 = time.time()
        for p in procs:
            p.start()
        got = dict(q.get(timeout=120) for _ in procs)
        for p in procs:
            p.join(60)
        assert not any(isinstance(v, str) for v in got.values()), got
        assert time.time() - t0 < 40                          # (the barrier's timeout is 60 s: nobody waited it out)
        for r in (0, 1):
            assert np.array_equal(got[r], np.arange(1000.0) * 3)
    finally:
        if os.path.exists(path):
            os.unlink(path)
