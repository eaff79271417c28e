"""Replica fan-out: INDEPENDENT units of work -- scan pairs to register, frames to fit -- spread over the GPUs of one
node with no communication between them ("Independent scan pairs: no communication ('replicas')").

The reference's unit of work is one pair: ``registration_gmmtree(source, target, ...)`` (src/python/hgmm/hgmm_gpu.py:
802-807) or ``registration_gmmreg(source, target, ...)`` (gmmreg_gpu/gmmreg.py:149-157), on one GPU.  Here a pool owns
one engine context per device (or several per device: a 40 k-point pair occupies a chip for a few milliseconds and
leaves most of it idle), each driven by its own thread; the library's calls drop the GIL, every C-ABI entry selects its
context's device itself (csrc/hgmm_ctx.h, HGMM_ENTER), so the threads never touch each other's state.  Jobs are handed
out dynamically (a pair that converges early frees its GPU for the next one); results come back in job order.

    from hgmm_amd.replicas import register_pairs
    results = register_pairs([(src0, tgt0), (src1, tgt1), ...], tree_level=3, maxiter=20)     # all visible GPUs

One process per GPU under an external launcher (torchrun) needs none of this: every rank simply registers its own pairs
on its own context (bench.py --mode pairs does exactly that and only meets the other ranks for the timing barrier).
"""
from __future__ import annotations

import ctypes
import queue
import threading

from ._native import Context, load_library, use_context


def device_count() -> int:
    n = ctypes.c_int(0)
    load_library().hgmm_device_count(ctypes.byref(n))
    return n.value


class ReplicaPool:
    """``contexts_per_device`` engine contexts on each of ``devices`` (default: every visible GPU), one worker thread
    per context.  ``map(fn, jobs)`` calls ``fn(ctx, job)`` for every job and returns the results in job order; the
    first exception of a worker is re-raised after the queue has drained."""

    def __init__(self, devices=None, contexts_per_device: int = 1, context_factory=Context):
        if devices is None:
            devices = list(range(max(device_count(), 1)))
        self.devices = [int(d) for d in devices for _ in range(max(int(contexts_per_device), 1))]
        if not self.devices:
            raise ValueError("ReplicaPool: no devices")
        self._factory = context_factory
        self._ctxs = [None] * len(self.devices)
        self._closed = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _context(self, i):
        if self._ctxs[i] is None:
            self._ctxs[i] = self._factory(self.devices[i])
        return self._ctxs[i]

    def map(self, fn, jobs):
        if self._closed:
            raise RuntimeError("ReplicaPool is closed")
        jobs = list(jobs)
        results = [None] * len(jobs)
        errors = []
        todo = queue.SimpleQueue()
        for item in enumerate(jobs):
            todo.put(item)

        def work(i):
            try:
                ctx = self._context(i)                       # created by the thread that uses it
                with use_context(ctx):                       # module-level mirror functions of this thread run on it
                    while not errors:
                        try:
                            k, job = todo.get_nowait()
                        except queue.Empty:
                            return
                        results[k] = fn(ctx, job)
            except BaseException as e:                       # noqa: BLE001 -- handed to the caller below
                errors.append((i, e))
                c = self._ctxs[i]                            # a context that failed mid-call is not handed out again
                self._ctxs[i] = None
                if c is not None:
                    try:
                        c.close()
                    except Exception:                        # noqa: BLE001 -- the original error is the one to report
                        pass

        threads = [threading.Thread(target=work, args=(i,), name="hgmm-replica-%d" % i, daemon=True)
                   for i in range(min(len(self.devices), max(len(jobs), 1)))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            i, e = errors[0]
            raise RuntimeError("replica %d (device %d) failed: %r" % (i, self.devices[i], e)) from e
        return results

    def close(self):
        self._closed = True
        for i, c in enumerate(self._ctxs):
            if c is not None:
                c.close()
                self._ctxs[i] = None


def register_pairs(pairs, devices=None, contexts_per_device: int = 1, maxiter: int = 20, tol: float = 1.0e-4,
                   method: str = "gmmtree", pool: ReplicaPool | None = None, batch: int = 1, **kargs):
    """Register every (source, target) pair of ``pairs`` -- arrays [N,3] or objects with ``.points`` -- with
    ``registration_gmmtree`` (``method='gmmtree'``, hgmm/hgmm_gpu.py:802-807; ``kargs`` = GMMTree's: tree_level,
    lambda_c, ls, sig2, ...) or ``registration_gmmreg`` (``method='gmmreg'``, gmmreg_gpu/gmmreg.py:149-157; it has no
    iteration budget or tolerance of its own, so ``maxiter`` / ``tol`` other than the defaults are refused), the pairs
    fanned out over the pool's contexts.  -> the reference's per-pair results, in the order of ``pairs``.

    ``batch`` > 1 (gmmtree only): every context takes ``batch`` pairs at a time through the SAME launches
    (``registration_gmmtree_batch``: the trees built as one forest, the targets registered together) -- a 40 k-point pair
    alone is a chain of ~350 small launches that leaves the chip idle; results are bitwise those of ``batch=1``.  Two
    contexts per device let one batch's uploads and 6 x 6 solves overlap the other's kernels."""
    pairs = list(pairs)
    if method == "gmmtree":
        from .hgmm.hgmm_gpu import registration_gmmtree, registration_gmmtree_batch

        if int(batch) > 1:
            def one(ctx, chunk):
                return registration_gmmtree_batch(chunk, maxiter=maxiter, tol=tol, ctx=ctx, **kargs)
        else:
            def one(ctx, pair):
                return registration_gmmtree(pair[0], pair[1], maxiter=maxiter, tol=tol, ctx=ctx, **kargs)
    elif method == "gmmreg":
        from .gmmreg_gpu.gmmreg import registration_gmmreg

        if int(batch) > 1:
            raise ValueError("batch > 1 is implemented for method='gmmtree' only")
        if maxiter != 20 or tol != 1.0e-4:
            raise ValueError("registration_gmmreg takes neither maxiter nor tol (its optimiser's settings live in "
                             "L2DistRegistration); leave them at their defaults")

        def one(ctx, pair):
            return registration_gmmreg(pair[0], pair[1], ctx=ctx, **kargs)
    else:
        raise ValueError("method must be 'gmmtree' or 'gmmreg', not %r" % (method,))
    own = pool is None
    pool = pool or ReplicaPool(devices, contexts_per_device)
    try:
        if method == "gmmtree" and int(batch) > 1:
            B = int(batch)
            chunks = [pairs[i:i + B] for i in range(0, len(pairs), B)]
            return [r for rs in pool.map(one, chunks) for r in rs]
        return pool.map(one, pairs)
    finally:
        if own:
            pool.close()


def fit_frames(frames, n_components=100, max_iter=30, tol=1.0e-4, cov_type="diag", devices=None,
               contexts_per_device: int = 1, pool: ReplicaPool | None = None):
    """One flat GMM per frame (``GMM_GPU_Base(n_components, max_iter, tol, cov_type).fit(frame)``,
    gmm_waymo/src/gmm.py:65-84), the frames fanned out over the pool's contexts.  -> the fitted models in frame order."""
    from .gmm_waymo.gmm import GMM_GPU_Base

    def one(ctx, frame):                                     # (the worker thread runs under use_context(ctx))
        return GMM_GPU_Base(n_components, max_iter, tol, cov_type).fit(frame)

    own = pool is None
    pool = pool or ReplicaPool(devices, contexts_per_device)
    try:
        return pool.map(one, frames)
    finally:
        if own:
            pool.close()
