"""Host side of the flat GMM EM path, shared by the two reference flavours.

``gmm_waymo/gmm_impl.py`` (variant "W": diag + spherical) and ``gmmreg_gpu/gmm_impl.py``
(variant "G": diag only) are thin, signature-compatible shells over these functions.
All arithmetic on points happens in the HIP kernels (csrc/flat_kernels.hip) through the
C ABI; this module only marshals small parameter arrays.
"""
from __future__ import annotations

import contextlib
import time

import numpy as np

from ._native import Context, DeviceArray, default_context


class DevicePoints:
    """A point cloud resident in HBM on one context (what ``cupy.asarray(X)`` was to the reference,
    gmm_waymo/src/gmm.py:73).  Pass it wherever the API takes ``X``.  Any number of them may be alive on a context
    (hgmm_points_*: each owns its device copy; using one binds it -- a pointer swap, nothing is uploaded again)."""

    def __init__(self, X, ctx: Context | None = None):
        self.ctx = ctx or default_context()
        X = np.asarray(X)
        if X.ndim != 2 or X.shape[1] != 3:
            raise ValueError("points must have shape [N,3], got %s" % (X.shape,))
        self.shape = X.shape
        self.dtype = np.dtype(np.float32)
        self._h = self.ctx.points_create(X)

    def __len__(self):
        return self.shape[0]

    def get(self):
        """The cloud back on the host, float32 [N,3] (``cupy.asnumpy``)."""
        if self._h is None:
            raise RuntimeError("this DevicePoints has been freed")
        return self.ctx.points_download(self._h)

    def bind(self):
        if self._h is None:
            raise RuntimeError("this DevicePoints has been freed")
        self.ctx.points_bind(self._h)
        return self.ctx

    def free(self):
        if self._h is not None:
            h, self._h = self._h, None
            self.ctx.points_destroy(h)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def asarray(X, ctx: Context | None = None):
    return X if isinstance(X, DevicePoints) else DevicePoints(X, ctx)


def _ctx_for(X) -> Context:
    """Returns the context holding X as its bound cloud: resident clouds are bound (nothing moves), host arrays are
    uploaded into the context's own cloud (like handing NumPy to the reference: a copy per call)."""
    if isinstance(X, DevicePoints):
        return X.bind()
    ctx = default_context()
    ctx.set_points(X)
    return ctx


def _host(a):
    return np.asarray(a.get() if isinstance(a, DeviceArray) else a)


def _param(a):
    """A parameter array as the E-step takes it: DeviceArrays stay where they are, anything else becomes NumPy."""
    return a if isinstance(a, DeviceArray) else np.asarray(a)


@contextlib.contextmanager
def timer(message, ctx=None):
    """gmm_impl.timer: synchronise, time, print (gmm_waymo/src/gmm_impl.py:43-50).  ``ctx``: the context whose stream
    is waited for (default: the process-wide one, like the reference's null stream)."""
    ctx = ctx or default_context()
    ctx.synchronize()
    start = time.time()
    yield
    ctx.synchronize()
    print('%s:  %f sec' % (message, time.time() - start))


def estimate_log_prob(X, inv_cov, means, cov_type):
    ctx = _ctx_for(X)
    return ctx.flat_log_prob(_host(inv_cov), _host(means), cov_type)


def e_step(X, inv_cov, means, weights, cov_type, variant):
    ctx = _ctx_for(X)
    # like the reference under CuPy nothing waits here: the mean is a device scalar, read when it is looked at
    # Parameters handed over as DeviceArrays (an earlier m_step's results, arithmetic on them) are used where they are.
    mean_lpn, log_resp, _, _ = ctx.flat_estep(_param(inv_cov), _param(means), _param(weights), cov_type, variant,
                                              lazy_mean=True)
    return mean_lpn, log_resp


def m_step(X, resp, cov_type, variant, centre_hint=None):
    """Array module in = array module out, like the reference (``xp = cupy.get_array_module(X)``, gmm_impl.py:91):
    responsibilities that live in HBM (a DeviceArray, e.g. ``log_resp.exp()`` of this module's e_step) give
    DeviceArrays -- nothing is downloaded, nothing waits; host responsibilities give NumPy arrays."""
    ctx = _ctx_for(X)
    return ctx.flat_mstep(resp, cov_type, variant, centre_hint, device_out=isinstance(resp, DeviceArray))


def train_gmm(X, max_iter, tol, means, covariances, weights, cov_type, variant):
    ctx = _ctx_for(X)
    inv, mu, w, cov, lls, converged = ctx.flat_train(max_iter, tol, _host(means), _host(covariances),
                                                     _host(weights), cov_type, variant)
    if not converged:
        print('Failed to converge. Increase max-iter or tol.')
    return inv, mu, w, cov, [np.float32(v) for v in lls]


def predict(X, inv_cov, means, weights, cov_type, variant):
    """Array module in = array module out (``xp.argmax``, gmm_impl.py:147-155): a resident cloud gives the labels as a
    DeviceArray (int32 in HBM; nothing is downloaded, nothing waits -- the reference's caller does the
    ``cupy.asnumpy``, run_gmm_static.py:49), a host array gives NumPy int64 like the reference under NumPy."""
    ctx = _ctx_for(X)
    lab = ctx.flat_predict(_param(inv_cov), _param(means), _param(weights), cov_type, variant)
    return lab if isinstance(X, DevicePoints) else lab.get().astype(np.int64)
