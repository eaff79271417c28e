"""world_size-2 CPU tests of the N>1 host path (no GPU): the unique-id bootstrap over gloo and
over plain TCP, the shard partition, and the algebra the library relies on -- centred sufficient
statistics of shards add up to the statistics of the whole cloud and give the oracle's M-step."""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest

from oracle import flat_em


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def exchange_bytes_torch(rank, world, payload):
    """A caller-provided transport for the unique id: torch.distributed's rendezvous (gloo, CPU).  Lives in the
    tests, not in the product package (north_star: no PyTorch)."""
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    obj = [payload if rank == 0 else None]
    dist.broadcast_object_list(obj, src=0)
    return obj[0]


def _worker(rank, world, port, transport, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from hgmm_amd import parallel
    payload = bytes(range(128)) if rank == 0 else None
    if transport == "torch":
        got = parallel.broadcast_from_rank0(rank, world, payload, exchange=exchange_bytes_torch)
    else:
        got = parallel.broadcast_from_rank0(rank, world, payload, transport=transport)
    if transport == "tcp":
        # the agreement round bench.py uses before choosing an all-reduce backend
        flags = parallel.allgather_bytes_tcp(rank, world, b"ok%d" % rank)
        assert flags == [b"ok%d" % r for r in range(world)]
    extra = None
    if transport == "torch":
        # the barrier / max-over-ranks reduction pattern bench.py uses, on gloo
        import torch
        import torch.distributed as dist
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        extra = float(t[0])
        dist.barrier()
        dist.destroy_process_group()
    q.put((rank, got, extra))


@pytest.mark.parametrize("transport", ["tcp", "torch"])
def test_unique_id_bootstrap_world2(transport):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, transport, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == bytes(range(128))
    if transport == "torch":
        assert res[0][2] == res[1][2] == 2.0


def test_shard_bounds_cover_exactly_once():
    from hgmm_amd.parallel import shard_bounds
    for n in (1, 7, 1000, 1_000_003):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def centred_stats(X, inv, mu, w):
    """What flat_fused_kernel accumulates (s0, a, b about mu), in float64 via the oracle."""
    _, lr = flat_em.e_step(X, inv, mu, w, "diag", "W")
    r = np.exp(lr)
    d = X[:, None, :] - mu[None]
    return np.concatenate([r.sum(0)[:, None], (r[:, :, None] * d).sum(0), (r[:, :, None] * d * d).sum(0)], axis=1)


def test_sharded_statistics_allreduce_equals_single_rank():
    rs = np.random.RandomState(0)
    X = rs.rand(3000, 3)
    mu = X[rs.choice(3000, 20, replace=False)]
    inv = 1 / np.sqrt(0.02 + 0.05 * rs.rand(20, 3))
    w = np.ones(20) / 20
    from hgmm_amd.parallel import shard_bounds
    whole = centred_stats(X, inv, mu, w)
    parts = sum(centred_stats(X[a:b], inv, mu, w) for a, b in (shard_bounds(3000, r, 2) for r in range(2)))
    np.testing.assert_allclose(parts, whole, rtol=1e-12, atol=1e-14)
    # M-step from the (all-reduced) centred statistics == the reference M-step (flavour W)
    s0, a, b = whole[:, 0], whole[:, 1:4], whole[:, 4:7]
    nk = s0 + 1e-8
    sx = a + mu * s0[:, None]
    sxx = b + 2 * mu * a + mu * mu * s0[:, None]
    m = sx / nk[:, None]
    cov = sxx / nk[:, None] - m * m + 1e-6
    _, lr = flat_em.e_step(X, inv, mu, w, "diag", "W")
    o_w, o_mu, o_cov = flat_em.m_step(X, np.exp(lr), "diag", "W")
    np.testing.assert_allclose(m, o_mu, rtol=1e-10)
    np.testing.assert_allclose(cov, o_cov, rtol=1e-8)
    np.testing.assert_allclose(nk / 3000, o_w, rtol=1e-12)


def _kmeans_reloc_worker(rank, world, port, q):
    """One rank of a world-2 KMeans empty-cluster relocation, all-reduces over gloo."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from hgmm_amd.kmeans import relocate_empty_sharded
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    def allreduce(values, op):
        t = torch.tensor(np.asarray(values, dtype=np.float64))
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
        return t.numpy()

    rs = np.random.RandomState(4)
    X = rs.rand(400, 3)
    centres = np.array([[0.2, 0.2, 0.2], [0.8, 0.8, 0.8], [9, 9, 9], [-7, 3, 3], [0.5, 0.5, 0.5]])
    d = ((X[:, None, :] - centres[None]) ** 2).sum(-1)
    labels, dist2 = d.argmin(1).astype(np.int32), d.min(1)
    lo, hi = (0, 170) if rank == 0 else (170, 400)              # uneven shards
    sums, counts = np.zeros((5, 3)), np.zeros(5)
    np.add.at(sums, labels, X)
    counts += np.bincount(labels, minlength=5)
    relocate_empty_sharded(allreduce, rank, labels[lo:hi], dist2[lo:hi], X[lo:hi], sums, counts)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, sums, counts))


def test_world2_kmeans_relocation_matches_single_process():
    """Sharded relocation == scikit-learn's single-process rule (the globally farthest points move to
    the empty clusters), and both ranks end with identical statistics."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_kmeans_reloc_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
    (_, s0, c0), (_, s1, c1) = res
    assert np.array_equal(s0, s1) and np.array_equal(c0, c1)
    # single-process reference (oracle = scikit-learn's rule)
    from oracle import kmeans as okm
    rs = np.random.RandomState(4)
    X = rs.rand(400, 3)
    centres = np.array([[0.2, 0.2, 0.2], [0.8, 0.8, 0.8], [9, 9, 9], [-7, 3, 3], [0.5, 0.5, 0.5]])
    labels, _ = okm.assign(X, centres)
    sums, counts = np.zeros((5, 3)), np.bincount(labels, minlength=5).astype(np.float64)
    np.add.at(sums, labels, X)
    okm._relocate_empty(X, centres, sums, counts, labels)
    assert np.array_equal(c0, counts)
    # the same two points move; which of the two empty clusters gets which is argpartition's order
    moved = {tuple(np.round(s0[j], 12)) for j in (2, 3)}
    assert moved == {tuple(np.round(sums[j], 12)) for j in (2, 3)}
    np.testing.assert_allclose(s0[[0, 1, 4]], sums[[0, 1, 4]], atol=1e-12)


def test_product_package_is_torch_free():
    """north_star: host code is NumPy + ctypes; PyTorch may start the ranks, it is never imported by the package."""
    import hgmm_amd
    root = os.path.dirname(os.path.abspath(hgmm_amd.__file__))
    pkg = os.path.realpath(os.path.join(root, os.pardir, "gpu-accelerated-point-cloud-registration-using-hierarchical-gmm_amd"))
    hits = []
    for base in {os.path.realpath(root), pkg}:
        for d, _, files in os.walk(base):
            for f in files:
                if f.endswith(".py"):
                    for n, line in enumerate(open(os.path.join(d, f)), 1):
                        t = line.strip()
                        if t.startswith(("import torch", "from torch")):
                            hits.append("%s:%d" % (os.path.join(d, f), n))
    assert not hits, hits


# ---- replica mode: TCP star between ranks that share no communicator, thread pool over contexts -------------------
def _tcp_group_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from hgmm_amd import parallel
    import time
    g = parallel.TcpGroup(rank, world)
    got = []
    for rnd in range(50):                                   # many rounds over the SAME connections
        got.append(g.allgather(("r%d-%d" % (rank, rnd)).encode() * (1 + rank)))
    time.sleep(0.05 * rank)                                 # ranks arrive at the barrier at different times
    g.barrier()
    t_after = time.time()
    mx = g.allgather_f64([float(rank), 10.0 - rank])
    g.barrier()
    g.close()
    q.put((rank, got, t_after, mx))


def test_tcp_group_allgather_barrier_world3():
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tcp_group_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = {}
    for _ in procs:
        rank, got, t_after, mx = q.get(timeout=60)
        out[rank] = (got, t_after, mx)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank in range(world):
        got, _, mx = out[rank]
        for rnd, row in enumerate(got):
            assert row == [("r%d-%d" % (r, rnd)).encode() * (1 + r) for r in range(world)]
        assert mx.shape == (world, 2) and list(mx[:, 0]) == [0.0, 1.0, 2.0] and mx[:, 1].max() == 10.0
    # nobody left the barrier before the slowest rank (0.1 s late) entered it
    ts = [out[r][1] for r in range(world)]
    assert max(ts) - min(ts) < 0.05


def test_replica_pool_fans_jobs_out_over_contexts_in_threads():
    """hgmm_amd.replicas.ReplicaPool without a GPU: one stand-in context per 'device', one thread each; results in job
    order, every job on exactly one context, module-level default_context() of a worker thread = its own context."""
    import threading
    import hgmm_amd
    from hgmm_amd.replicas import ReplicaPool

    class Fake:
        def __init__(self, device):
            self.device, self.h, self.jobs, self.busy = device, 1, [], threading.Lock()

        def close(self):
            self.h = None

    made = []

    def factory(device):
        made.append(Fake(device))
        return made[-1]

    def fn(ctx, job):
        assert hgmm_amd.default_context() is ctx            # use_context: the thread's default IS its replica's context
        assert ctx.busy.acquire(blocking=False)             # a context is never driven by two threads at once
        try:
            ctx.jobs.append(job)
            import time
            time.sleep(0.002)
            return job * job
        finally:
            ctx.busy.release()

    with ReplicaPool(devices=[0, 1, 2], contexts_per_device=2, context_factory=factory) as pool:
        assert pool.devices == [0, 0, 1, 1, 2, 2]
        res = pool.map(fn, range(40))
        assert res == [j * j for j in range(40)]
        assert sorted(j for c in made for j in c.jobs) == list(range(40))
        assert len(made) == 6 and sum(1 for c in made if c.jobs) >= 2
        res2 = pool.map(fn, [5])                            # fewer jobs than contexts
        assert res2 == [25]
    assert all(c.h is None for c in made)
    # a failing job stops the pool and surfaces
    with ReplicaPool(devices=[0, 1], context_factory=factory) as pool:
        def bad(ctx, job):
            if job == 3:
                raise ValueError("job 3")
            return job
        try:
            pool.map(bad, range(8))
            raise AssertionError("no error")
        except RuntimeError as e:
            assert "job 3" in repr(e.__cause__)
    # outside a pool the thread-local override is gone
    assert getattr(hgmm_amd._native._thread_ctx, "ctx", None) is None
