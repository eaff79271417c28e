"""Drop-in for the reference's ``src/python/gmmreg_gpu/gmm_impl.py`` (flavour "G": diag only,
KMeans initialisation, no eps inside log(weights), clipped covariance)."""
import numpy as np

from .. import _flat
from .._flat import asarray, timer, DevicePoints  # noqa: F401

eps = 1e-8
VARIANT = "G"


def init_gmm_params(X, k):
    """KMeans(k, random_state=1, max_iter=50, n_init=1) centres + uniform weights
    (reference gmm_impl.py:18-24).  Host-side, scikit-learn like the reference."""
    from sklearn.cluster import KMeans
    kmeans = KMeans(n_clusters=k, random_state=1, max_iter=50, n_init=1).fit(np.asarray(X))
    return kmeans.cluster_centers_, np.ones((k)) / k


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
