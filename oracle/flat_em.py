"""CPU oracle for the flat (non-hierarchical) GMM EM hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package may import this
module: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / the timed CPU baseline.

This is a NumPy restatement (no code copied) of the reference's two flat EM
variants; every function cites the reference file:line it follows (paths are
relative to the upstream repository root):

* variant ``"W"`` = ``src/python/gmm_waymo/src/gmm_impl.py``  (diag + spherical)
* variant ``"G"`` = ``src/python/gmmreg_gpu/gmm_impl.py``     (diag only)

The arithmetic is dtype-generic exactly like the reference (which dispatches on
the array module and keeps whatever dtype it is handed): called with float32
arrays it reproduces the reference's fp32 op sequence (same GEMM shapes, same
temporaries, same expanded quadratic form, same non-max-shifted log-sum-exp);
called with float64 arrays it is the high-precision yard-stick the GPU kernels
are compared against.

Pinning: ``tests/golden/flat_*.npz`` were produced by importing the reference
itself (``tools/gen_golden.py``) and ``tests/test_oracle_golden.py`` checks this
module against them, so parity is pinned to the reference's own outputs.
"""
from __future__ import annotations

import numpy as np

EPS = 1e-8          # gmm_waymo/src/gmm_impl.py:15, gmmreg_gpu/gmm_impl.py:16
REG_COVAR = 1e-6    # gmm_waymo/src/gmm_impl.py:81 (estimate_covariance default)

VARIANTS = ("W", "G")


def _log2pi_f32():
    # Both variants cast log(2*pi) to float32 before use
    # (gmm_waymo gmm_impl.py:65,78; gmmreg_gpu gmm_impl.py:43).
    return np.log(2 * np.pi).astype(np.float32)


def log_gauss_diag(X, inv_std, mu):
    """Per-pair log N(x_i; mu_j, diag) in the reference's *expanded* form.

    Follows gmm_waymo gmm_impl.py:67-78 == gmmreg_gpu gmm_impl.py:36-43.
    ``inv_std`` is the reference's ``inv_cov`` (= 1/sigma, shape [J,3]).
    """
    d = X.shape[1]
    half_log_det = np.sum(np.log(inv_std + EPS), axis=1)
    prec = inv_std ** 2
    quad = (np.sum(mu ** 2 * prec, 1)
            - 2 * np.dot(X, (mu * prec).T)
            + np.dot(X ** 2, prec.T))
    return -0.5 * (d * _log2pi_f32() + quad) + half_log_det


def log_gauss_spherical(X, inv_std, mu):
    """Spherical covariance; follows gmm_waymo gmm_impl.py:53-65 (+row_norms 18-24)."""
    d = X.shape[1]
    half_log_det = d * np.log(inv_std + EPS)
    prec = inv_std ** 2
    x2 = np.einsum('ij,ij->i', X, X)
    quad = (np.sum(mu ** 2, 1) * prec
            - 2 * np.dot(X, mu.T * prec)
            + np.outer(x2, prec))
    return -0.5 * (d * _log2pi_f32() + quad) + half_log_det


def _log_gauss(X, inv_std, mu, cov_type):
    if cov_type == 'diag':
        return log_gauss_diag(X, inv_std, mu)
    if cov_type == 'spherical':
        return log_gauss_spherical(X, inv_std, mu)
    raise ValueError("cov_type must be 'diag' or 'spherical'")


def _log_weights(w, variant):
    # W adds eps inside the log (gmm_impl.py:109,111); G does not (gmm_impl.py:57-58).
    return np.log(w + EPS) if variant == "W" else np.log(w)


def e_step(X, inv_std, mu, w, cov_type='diag', variant='W'):
    """Returns (mean_i log-normaliser, log_resp[N,J]).

    gmm_waymo gmm_impl.py:105-116 / gmmreg_gpu gmm_impl.py:55-61.  Note the
    normaliser is ``log(sum(exp(wlp)) + eps)`` with NO max-shift, so rows do not
    sum to one when every exponential is tiny.
    """
    if variant == "G" and cov_type != 'diag':
        raise ValueError("variant G is diag-only")
    wlp = _log_gauss(X, inv_std, mu, cov_type) + _log_weights(w, variant)
    lpn = np.log(np.sum(np.exp(wlp), axis=1) + EPS)
    return np.mean(lpn), wlp - lpn[:, None]


def e_step_full(X, inv_std, mu, w, cov_type='diag', variant='W'):
    """Like :func:`e_step` but also returns the per-point normaliser and argmax."""
    wlp = _log_gauss(X, inv_std, mu, cov_type) + _log_weights(w, variant)
    lpn = np.log(np.sum(np.exp(wlp), axis=1) + EPS)
    return np.mean(lpn), wlp - lpn[:, None], lpn, wlp.argmax(axis=1)


def m_step(X, resp, cov_type='diag', variant='W'):
    """Returns (weights, means, covariances).

    W: gmm_waymo gmm_impl.py:90-103 with estimate_covariance 81-88
       (nk = sum r + eps; cov = E[x^2] - 2 mu E[x] + mu^2 + 1e-6; spherical = mean over d).
    G: gmmreg_gpu gmm_impl.py:46-52 (nk = sum r; divisions by nk + eps; clip at 0).
    """
    n = len(X)
    if variant == "W":
        nk = np.sum(resp, axis=0) + EPS
        mu = np.dot(resp.T, X) / nk[:, None]
        ex2 = np.dot(resp.T, X * X) / nk[:, None]
        mu2 = mu ** 2
        mu_ex = mu * np.dot(resp.T, X) / nk[:, None]
        cov = ex2 - 2 * mu_ex + mu2 + REG_COVAR
        if cov_type == 'spherical':
            cov = np.mean(cov, axis=1)
        return nk / n, mu, cov
    nk = np.sum(resp, axis=0)
    mu = np.dot(resp.T, X) / (nk[:, None] + EPS)
    ex2 = np.dot(resp.T, X * X) / (nk[:, None] + EPS)
    cov = np.clip(ex2 - mu ** 2, 0.0, None)
    return nk / n, mu, cov


def inv_std_from_cov(cov, variant='W', initial=False):
    """The reference's ``inv_cov`` update.

    initial: 1/sqrt(cov)                      (W gmm_impl.py:122, G gmm_impl.py:67)
    W loop : 1/(sqrt(cov + 1e-6) + eps)       (gmm_impl.py:134)
    G loop : 1/(sqrt(cov) + eps)              (gmm_impl.py:74)
    """
    if initial:
        return 1 / np.sqrt(cov)
    if variant == "W":
        return 1 / (np.sqrt(cov + 1e-6) + EPS)
    return 1 / (np.sqrt(cov) + EPS)


def train(X, max_iter, tol, mu, cov, w, cov_type='diag', variant='W'):
    """EM loop; returns (inv_std, mu, w, cov, lls, converged).

    gmm_waymo gmm_impl.py:118-145 / gmmreg_gpu gmm_impl.py:63-85.  ``lls[k]`` is the
    mean log-normaliser evaluated with the parameters *before* the k-th M-step; the
    loop stops after the M-step of the first iteration whose |delta lls| < tol.
    """
    prev = -np.inf
    converged = False
    inv_std = inv_std_from_cov(cov, variant, initial=True)
    lls = []
    for _ in range(max_iter):
        ll, log_resp = e_step(X, inv_std, mu, w, cov_type, variant)
        lls.append(ll)
        w, mu, cov = m_step(X, np.exp(log_resp), cov_type, variant)
        inv_std = inv_std_from_cov(cov, variant)
        change = ll - prev
        prev = ll
        if abs(change) < tol:
            converged = True
            break
    return inv_std, mu, w, cov, lls, converged


def predict(X, inv_std, mu, w, cov_type='diag', variant='W'):
    """Hard labels; gmm_waymo gmm_impl.py:147-155 / gmmreg_gpu gmm_impl.py:88-91."""
    return (_log_gauss(X, inv_std, mu, cov_type) + _log_weights(w, variant)).argmax(axis=1)


# ---------------------------------------------------------------------------
# Helpers used by the parity tests (not part of the reference's surface).
# ---------------------------------------------------------------------------

def seeded_init(X, k, seed=0, cov_type='diag', dtype=np.float32):
    """Deterministic initialisation used by the benchmarks/tests:
    means = k distinct points, cov = 0.1 (reference constant gmm_impl.py:37,39), w = 1/k."""
    rs = np.random.RandomState(seed)
    idx = rs.choice(len(X), k, replace=False)
    mu = np.ascontiguousarray(X[idx]).astype(dtype)
    w = (np.ones(k) / k).astype(dtype)
    if cov_type == 'diag':
        cov = (0.1 * np.ones((k, X.shape[1]))).astype(dtype)
    else:
        cov = (0.1 * np.ones((k,))).astype(dtype)
    return mu, w, cov


def near_tie_mask(log_resp, tol=1e-5):
    """Rows whose top-2 responsibilities differ by < tol (genuine near-ties where a
    label flip between two correct implementations is legitimate)."""
    r = np.exp(np.asarray(log_resp, dtype=np.float64))
    part = np.partition(r, -2, axis=1)
    return (part[:, -1] - part[:, -2]) < tol
