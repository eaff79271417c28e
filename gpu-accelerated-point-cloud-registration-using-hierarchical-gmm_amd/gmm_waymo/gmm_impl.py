"""Drop-in for the reference's ``src/python/gmm_waymo/src/gmm_impl.py`` (flavour "W").

Same function names, argument order and return tuples; the N x J arithmetic runs in the
hand-written HIP kernels instead of CuPy/NumPy array expressions.  ``X`` may be a host
``numpy.ndarray`` [N,3] (uploaded on every call, like handing NumPy to the reference) or a
``DevicePoints`` from :func:`asarray` (uploaded once -- what ``cupy.asarray`` was).
Large results (``log_resp``) come back as ``DeviceArray`` (``.get()`` / ``np.asarray`` to
download), exactly where the reference returned CuPy arrays.
"""
import numpy as np

from .. import _flat
from .._flat import asarray, timer, DevicePoints  # noqa: F401  (re-exported)

float32 = np.float32
eps = 1e-8            # reference gmm_impl.py:15
VARIANT = "W"


def init_gmm_params(X, k, cov_type='diag'):
    """weights = 1/k, means = k*3 random *scalars* drawn from X.flatten(), covs = 0.1
    (reference gmm_impl.py:26-41; the RNG stays on the host)."""
    Xh = np.asarray(X) if not isinstance(X, DevicePoints) else None
    if Xh is None:
        raise TypeError("init_gmm_params needs the host array (it samples from X on the host)")
    weights = np.ones(k, dtype=np.float32) / k
    means = np.random.choice(Xh.flatten(), (k, Xh.shape[1]))
    if cov_type == 'diag':
        covs = 0.1 * np.ones((k, Xh.shape[1]), dtype=np.float32)
    elif cov_type == 'spherical':
        covs = 0.1 * np.ones((k,), dtype=np.float32)
    else:
        raise ValueError("cov_type must be 'diag' or 'spherical'")
    return means.astype(np.float32), weights, covs


def estimate_log_prob(X, inv_cov, means):
    """-> log N(x_i; mu_j, diag) [N,J] DeviceArray.  Reference gmm_impl.py:67-78."""
    return _flat.estimate_log_prob(X, inv_cov, means, 'diag')


def estimate_log_prob_spherical(X, inv_cov, means):
    """Reference gmm_impl.py:53-65."""
    return _flat.estimate_log_prob(X, inv_cov, means, 'spherical')


def row_norms(X, squared=False):
    """Host helper kept for API parity (reference gmm_impl.py:18-24)."""
    n = np.einsum('ij,ij->i', np.asarray(X), np.asarray(X))
    return n if squared else np.sqrt(n)


def e_step(X, inv_cov, means, weights, cov_type='diag'):
    """-> (mean log-normaliser, log_resp[N,J] DeviceArray).  Reference gmm_impl.py:105-116."""
    return _flat.e_step(X, inv_cov, means, weights, cov_type, VARIANT)


def m_step(X, resp, cov_type='diag', centre_hint=None):
    """-> (weights, means, covariances).  Reference gmm_impl.py:90-103.  ``resp`` may be a
    host array, a DeviceArray, or ``log_resp.exp()`` (lazy; the exp is fused)."""
    return _flat.m_step(X, resp, cov_type, VARIANT, centre_hint)


def train_gmm(X, max_iter, tol, means, covariances, weights, cov_type='diag'):
    """-> (inv_cov, means, weights, covariances, log_ll).  Reference gmm_impl.py:118-145.
    The whole loop is device-resident (fused E+M kernels, one read-back at the end)."""
    return _flat.train_gmm(X, max_iter, tol, means, covariances, weights, cov_type, VARIANT)


def predict(X, inv_cov, means, weights, cov_type='diag'):
    """-> labels[N] int64.  Reference gmm_impl.py:147-155."""
    return _flat.predict(X, inv_cov, means, weights, cov_type, VARIANT)
