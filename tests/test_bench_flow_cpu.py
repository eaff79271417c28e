"""Control-flow tests of bench.py's N > 1 path on CPU (the real engine needs a GPU; kernels are not run
here): a stand-in engine context whose collectives go through gloo.

  * two ranks spawned the way torch.distributed.run would (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*):
    the plain-TCP unique-id bootstrap, barrier / max-over-ranks block timing, exactly one well-formed JSON
    line from rank 0 (with `allreduce_us` for N > 1);
  * the same through bench.py's OWN launcher (`launch_ranks`, what `python bench.py --gpus N` without a
    launcher does), including the exit-code path of a failing rank and the RCCL -> host-backend retry."""
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

    def comm_init_host(self, world, rank, name):
        self.world, self.rank, self.hostcomm = world, rank, name

    def comm_init_ipc(self, world, rank, name):
        self.world, self.rank, self.ipc = world, rank, name

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
        # the stand-in's collectives ride on gloo (the product's ride on RCCL inside the library)
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ["MASTER_PORT"] = str(int(os.environ["MASTER_PORT"]) + 5)
            dist.init_process_group(backend="gloo", rank=int(os.environ["RANK"]),
                                    world_size=int(os.environ["WORLD_SIZE"]))
        t = torch.tensor(np.asarray(values, dtype=np.float64))
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
        return t.numpy()

    def flat_train_begin(self, tol, mu, cov, w, cov_type, variant, lls_capacity):
        self.J = len(mu)
        self.n_it = 0

    def flat_train_step(self, iters):
        self.n_it += iters

    def flat_train_end(self):
        J = self.J
        z = np.zeros((J, 3), np.float32)
        return z, z, np.zeros(J, np.float32), z, np.zeros(self.n_it, np.float32), False, self.n_it

    def flat_train(self, max_iter, tol, mu, cov, w, cov_type="diag", variant="W"):
        z = np.zeros_like(np.asarray(mu, dtype=np.float32))
        return z, z, np.zeros(len(z), np.float32), z, np.zeros(max_iter, np.float32), False

    def empty(self, shape, dtype=np.float32):
        class _Arr:
            def free(self_inner):
                pass
        return _Arr()

    def flat_estep(self, inv, mu, w, cov_type="diag", variant="W", out=None, **kw):
        return 0.0, out, None, None

    def tree_build(self, L, ls, ld, init_mu, sig2, max_iters_per_level=1000, q_capacity=None, want_leaf=True):
        T = len(init_mu)
        self._coll = getattr(self, "_coll", 0) + 1 + 2 * 4 * 5         # 3 iterations + 2 ahead on each of 4 levels
        self._surplus = getattr(self, "_surplus", 0) + 2 * 4
        return np.ones(T), np.ones((T, 3)), np.ones((T, 3, 3)), None, np.full(L, 3, np.int32), np.zeros(3 * L)

    def comm_stats(self):
        return getattr(self, "_coll", 0), getattr(self, "_surplus", 0)

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
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "3", "--warmup", "1", "--min-time", "0"]
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
    # the compact per-leg summary closes the line (a reader that keeps only the line's tail still has it)
    assert list(d)[-1] == "summary" and d["summary"]["it_per_s"] > 0 and "legs" in d
    # an N > 1 line must not read as "unmeasured": roofline (rank 0's E-step launch) and cpu_baseline are objects
    assert d["value"] > 0 and d["it_per_s_per_gpu"] * 2 == d["value"]
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["frac"] > 0 and "cold_frac" in d["roofline"]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    assert d["rank_consistency"]["identical_model_on_all_ranks"] is True
    assert "workload" in d["config"] and "RCCL" in d["config"]["collective"] and d["config"]["collective_requested"] == "auto"
    assert d["allreduce_us"] == 500.0 and d["timing"]["blocks"] >= 3 and d["timing"]["steps_per_block"] == 3
    # --collective auto: the same joint fit once more over the one-shot peer exchange, beside the RCCL headline
    px = d["peer_exchange"]
    assert "peer exchange" in px["collective"] and px["value"] > 0 and px["identical_model_on_all_ranks"] is True
    assert px["model_bitwise_equal_to_the_headline_fit"] is True
    # the sharded HGMM leg: what the look-ahead behind a device-side stop costs under a communicator
    st = d["sharded_tree"]
    assert st["level_iterations"] == [3, 3, 3, 3] and st["collectives"] == 41 and st["identical_tree_on_all_ranks"] is True
    assert d["surplus_collectives"] == st["surplus_collectives"] == 16 and st["surplus_level_iterations"] == 8
    assert "torch" not in open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                            "bench.py")).read().split('"""', 2)[2].replace("torch.distributed.run", "")


HELPER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bench_fake_rank.py")


def _run_launcher(n, mode, extra_args=()):
    """python -c 'bench.launch_ranks / self_launch' in a child so that rank 0's inherited stdout can be captured."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, types; sys.path.insert(0, %r); import bench\n"
            "args = types.SimpleNamespace(gpus=%d)\n"
            "argv = ['--gpus', '%d', '--steps', '2', '--warmup', '1', '--min-time', '0'] + %r\n"
            "bench.__file__ = %r\n"
            "import os\n"
            "orig = bench.launch_ranks\n"
            "bench.launch_ranks = lambda n, argv, script=None, **kw: orig(n, argv, script=%r, **kw)\n"
            "sys.exit(bench.self_launch(args, argv) if %r == 'self' else bench.launch_ranks(%d, argv))\n"
            % (root, n, n, list(extra_args), HELPER, HELPER, mode, n))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)


def test_self_launch_two_ranks():
    r = _run_launcher(2, "self")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and "allreduce_us" in d


def test_self_launch_reports_a_failing_rank():
    r = _run_launcher(2, "plain", ["--skip", "FAIL_RANK_1"])
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])


def test_ranks_of_an_external_launcher_fall_back_in_place_when_rccl_is_unavailable():
    """Started by something that is not bench.py's own launcher (torchrun in the driver's N > 1 runs): no restart is
    possible, every rank moves on to the next backend -- host shared memory, the one that needs nothing from the GPUs'
    interconnect -- and the line is still produced; the peer exchange is timed beside it as the side leg."""
    r = _run_launcher(2, "plain", ["--skip", "RCCL_BROKEN"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert "rccl" in d["config"]["fallback"] and "host shared memory" in d["config"]["collective"]
    assert "peer exchange" in d["peer_exchange"]["collective"] and d["peer_exchange"]["value"] > 0


def test_fallback_chain_keeps_the_line_when_the_peer_exchange_is_unavailable_too():
    """Neither RCCL nor the peer exchange: the headline runs on host shared memory and the side leg reports that the
    exchange could not be set up -- under bench.py's own launcher as under an external one."""
    for mode in ("self", "plain"):
        r = _run_launcher(2, mode, ["--skip", "RCCL_AND_IPC_BROKEN"])
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
        assert len(lines) == 1, r.stdout
        d = json.loads(lines[0])
        assert "rccl" in d["config"]["fallback"] and "host shared memory" in d["config"]["collective"]
        assert "error" in d["peer_exchange"]


def test_explicit_collective_is_used_alone():
    r = _run_launcher(2, "plain", ["--collective", "ipc"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][0])
    assert "peer exchange" in d["config"]["collective"] and "fallback" not in d["config"] and "peer_exchange" not in d


def test_partial_rccl_failure_is_agreed_on_collectively():
    """communicator creation failing on ONE rank only must not leave the ranks on different backends --
    the ok flags are all-gathered over TCP first, then every rank falls back together."""
    r = _run_launcher(2, "plain", ["--skip", "RCCL_BROKEN_ON_RANK_1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert "fallback" in d["config"] and "host shared memory" in d["config"]["collective"]
    assert d["rank_consistency"]["identical_model_on_all_ranks"] is True


def test_a_hanging_backend_setup_times_out_and_falls_back():
    """a rank that blocks inside the collective communicator set-up (its peer failed there) must not wait
    for ever: the attempt is abandoned after HGMM_BENCH_ATTACH_TIMEOUT and the agreement round moves everybody on."""
    os.environ["HGMM_BENCH_ATTACH_TIMEOUT"] = "3"
    try:
        r = _run_launcher(2, "plain", ["--skip", "RCCL_HANGS_ON_RANK_0"])
    finally:
        os.environ.pop("HGMM_BENCH_ATTACH_TIMEOUT", None)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][0])
    assert "rccl" in d["config"]["fallback"] and "host shared memory" in d["config"]["collective"]


def test_pairs_mode_two_ranks_without_a_communicator():
    """--mode pairs: every rank registers its own scan pairs, the ranks only meet over the TCP star for the timing
    barrier and the max-over-ranks block time; one line from rank 0, no collective in it."""
    for mode in ("self", "plain"):
        r = _run_launcher(2, mode, ["--mode", "pairs"])
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
        assert len(lines) == 1, r.stdout
        d = json.loads(lines[0])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                  "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in d
        assert d["mode"] == "pairs" and d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0
        assert d["pairs_per_s_per_gpu"] * 2 == d["value"] and d["scaling"] == "weak"
        assert d["registration_iterations_per_pair"] == 7.0 and d["accuracy"]["ok"] is True
        assert "collective" not in d["config"] and "allreduce_us" not in d
        assert d["timing"]["blocks"] >= 3
        # default: batches of pairs through the same launches; value counts pairs, not steps
        B, C = d["config"]["batch"], d["config"]["contexts_per_gpu"]
        assert B > 1 and d["config"]["pairs_per_gpu_per_step"] == B * C
        assert abs(d["value"] * d["timing"]["median_block_ms"] * 1e-3 - 2 * C * d["steps"] * B) < 1e-6 * d["value"]
    r = _run_launcher(1, "plain", ["--mode", "pairs", "--batch", "1", "--no-cpu-baseline"])         # round 5's path: one call per pair
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][0])
    assert d["config"]["batch"] == 1 and d["config"]["pairs_per_gpu_per_step"] == d["config"]["contexts_per_gpu"]


def test_pairs_mode_single_rank_carries_a_cpu_baseline():
    r = _run_launcher(1, "plain", ["--mode", "pairs"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][0])
    assert d["n_gpus"] == 1 and d["cpu_baseline"]["kind"] == "port" and d["mode"] == "pairs"


def test_stdout_carries_the_json_line_only():
    """bench.py's stdout guard: whatever libraries print to file descriptor 1 while the bench runs (RCCL's version
    banner through C stdio, flushed at exit) lands on stderr; the restored stdout carries the one line."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, os, ctypes; sys.path.insert(0, %r); import bench\n"
            "libc = ctypes.CDLL(None)\n"
            "g = bench._StdoutGuard()\n"
            "libc.puts(b'banner through C stdio (buffered until exit)')\n"
            "os.write(1, b'raw write to fd 1\\n')\n"
            "g.restore()\n"
            "print('{\"the\": \"line\"}'); sys.stdout.flush()\n"
            "g2 = bench._StdoutGuard()\n"
            "libc.puts(b'teardown chatter')\n" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == '{"the": "line"}', r.stdout
    assert "banner through C stdio" in r.stderr and "raw write to fd 1" in r.stderr and "teardown chatter" in r.stderr
