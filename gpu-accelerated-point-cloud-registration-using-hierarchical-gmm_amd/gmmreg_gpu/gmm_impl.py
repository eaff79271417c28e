"""Drop-in for the reference's ``src/python/gmmreg_gpu/gmm_impl.py`` (flavour "G": diag only,
KMeans initialisation, no eps inside log(weights), clipped covariance)."""
import numpy as np

from .. import _flat
from .._flat import asarray, timer, DevicePoints  # noqa: F401

eps = 1e-8
VARIANT = "G"


def init_gmm_params(X, k):
    """KMeans(k, random_state=1, max_iter=50, n_init=1) centres + uniform weights
    (reference gmm_impl.py:18-24).  The reference calls scikit-learn on the host; here the same
    algorithm (same random stream, same seeds, same stop rule) runs on the device
    (``hgmm_amd.kmeans``, ``csrc/kmeans_kernels.hip``)."""
    from ..kmeans import kmeans_centres
    return kmeans_centres(X, k, random_state=1, max_iter=50), np.ones((k)) / k


def estimate_log_prob(X, inv_cov, means):
    """Reference gmmreg_gpu/gmm_impl.py:36-43."""
    return _flat.estimate_log_prob(X, inv_cov, means, 'diag')


def e_step(X, inv_cov, means, weights):
    return _flat.e_step(X, inv_cov, means, weights, 'diag', VARIANT)


def m_step(X, resp, centre_hint=None):
    return _flat.m_step(X, resp, 'diag', VARIANT, centre_hint)


def train_gmm(X, max_iter, tol, means, covariances, weights=None):
    if weights is None:
        weights = np.ones(len(means), dtype=np.float32) / len(means)
    return _flat.train_gmm(X, max_iter, tol, means, covariances, weights, 'diag', VARIANT)


def predict(X, inv_cov, means, weights):
    return _flat.predict(X, inv_cov, means, weights, 'diag', VARIANT)
