"""Control-flow test of bench.py's N > 1 path on CPU: two ranks (spawned like torch.distributed.run
would: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), a stand-in engine context whose collectives go
through gloo.  Checks the rendezvous bootstrap, barrier / max-over-ranks timing and that exactly
rank 0 prints one well-formed JSON line.  (The real engine needs a GPU; kernels are not run here.)"""
import json
import multiprocessing as mp
import os
import socket
import sys

import numpy as np


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class FakeContext:
    """Implements exactly the Context surface bench.py touches."""

    def __init__(self, device_id=0):
        self.device_id = device_id
        self.n_it = 0

    @staticmethod
    def comm_unique_id():
        return bytes(range(128))

    def comm_init(self, world, rank, uid):
        assert uid == bytes(range(128))
        self.world, self.rank = world, rank

    def comm_destroy(self):
        pass

    def close(self):
        pass

    def device_info(self):
        return {"name": "fake", "compute_units": 256, "hbm_bytes": 1}

    def set_points(self, X):
        self.n = len(X)
        return self

    def synchronize(self):
        pass

    def allreduce(self, values, op="sum"):
        import torch
        import torch.distributed as dist
        t = torch.tensor(np.asarray(values, dtype=np.float64))
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
        return t.numpy()

    def flat_train_begin(self, tol, mu, cov, w, cov_type, variant, lls_capacity):
        self.J = len(mu)

    def flat_train_step(self, iters):
        self.n_it += iters

    def flat_train_end(self):
        J = self.J
        z = np.zeros((J, 3), np.float32)
        return z, z, np.zeros(J, np.float32), z, np.zeros(self.n_it, np.float32), False, self.n_it

    def profile_reset(self):
        pass

    def profile_enable(self, on=True):
        pass

    def profile_get(self, kernel):
        return 1.0, 2


def _rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import io
    import contextlib
    import hgmm_amd
    import bench
    hgmm_amd.Context = FakeContext
    bench.N_POINTS = 2000                       # keep the synthetic frames tiny
    bench.synth_frame = lambda seed, n=2000: np.random.RandomState(seed).rand(n, 3).astype(np.float32)
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "3", "--warmup", "1"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()
    q.put((rank, buf.getvalue()))


def test_bench_two_rank_flow():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        out = dict(q.get(timeout=120) for _ in procs)
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()
    for p in procs:
        assert p.exitcode == 0
    assert out[1].strip() == ""                         # only rank 0 prints
    lines = [l for l in out[0].splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["roofline"] is None and d["cpu_baseline"] is None
    assert "workload" in d["config"]
