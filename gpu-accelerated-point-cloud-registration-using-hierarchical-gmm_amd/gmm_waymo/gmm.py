"""Drop-in for the reference's ``src/python/gmm_waymo/src/gmm.py``: the sklearn-style feature
extractors that callers (run_gmm_static.py:35-49, run_gmm_waymo_gpu.py, gmmreg.py:71,84,144)
use -- ``init()``, ``compute(data)``, ``predict(data)``, ``fit(X)``.

Every class here runs on the MI355X engine; there is no CPU compute path in this package.
``GMM_CPU`` / ``GMM_CPU_Base`` are kept for API compatibility (3-tuple ``compute``,
reference gmm.py:103-149) and differ from ``GMM_GPU`` only in what they return.
"""
import abc

import numpy as np

from . import gmm_impl
from .gmm_impl import train_gmm, init_gmm_params, timer, predict, asarray  # noqa: F401


class Feature(abc.ABC):
    """Feature protocol (reference gmm.py:14-27)."""

    @abc.abstractmethod
    def init(self):
        pass

    @abc.abstractmethod
    def compute(self, data):
        return None

    def annealing(self):
        pass

    def __call__(self, data):
        return self.compute(data)


class GMM_GPU_Base:
    """fit/predict (reference gmm.py:65-101).  After ``fit``: ``means_``, ``covariances_``,
    ``weights_``, ``lls``, ``inv_covs``."""

    _label = 'GPU GMM TRAIN'
    _verbose = True

    def __init__(self, num_components, max_iter=30, tol=1e-4, cov_type='diag'):
        self.num_components = num_components
        self.max_iter = max_iter
        self.tol = tol
        self.cov_type = cov_type

    def _init_params(self, X):
        return init_gmm_params(X, self.num_components, cov_type=self.cov_type)

    def fit(self, X, init=None):
        """``init`` = optional explicit ``(means, weights, covs)`` (the reference always draws
        them with the host RNG; tests and benchmarks pass them in)."""
        X = np.asarray(X)
        means, weights, covs = init if init is not None else self._init_params(X)
        dev_X = asarray(X.astype(np.float32))
        with timer(self._label):
            inv, mu, w, cov, lls = train_gmm(dev_X, self.max_iter, self.tol,
                                             np.asarray(means, np.float32), np.asarray(covs, np.float32),
                                             np.asarray(weights, np.float32), cov_type=self.cov_type)
        self.means_, self.covariances_, self.weights_ = mu, cov, w
        self.lls, self.inv_covs = lls, inv
        if self._verbose and len(lls):
            print("\nLog Likelihood Min-Max:\n\n", np.min(lls), np.max(lls))
        return self

    def predict(self, X):
        X = np.asarray(X).astype(np.float32)
        return predict(X, self.inv_covs, self.means_, self.weights_, cov_type=self.cov_type)


class GMM_CPU_Base(GMM_GPU_Base):
    _label = 'CPU GMM TRAIN'
    _verbose = False


class GMM_GPU(Feature):
    """reference gmm.py:46-63; ``compute`` -> (means, weights, covariances, inv_covs)."""
    _base = GMM_GPU_Base

    def __init__(self, n_gmm_components=100, max_iter=30, tol=1e-4, cov_type='diag'):
        self._n_gmm_components = n_gmm_components
        self.max_iter = max_iter
        self.tol = tol
        self.cov_type = cov_type

    def init(self):
        self._clf = self._base(self._n_gmm_components, max_iter=self.max_iter, tol=self.tol,
                               cov_type=self.cov_type)

    def compute(self, data):
        self._clf.fit(data)
        return self._clf.means_, self._clf.weights_, self._clf.covariances_, self._clf.inv_covs

    def predict(self, data):
        return self._clf.predict(data)


class GMM_CPU(GMM_GPU):
    """reference gmm.py:103-118; ``compute`` -> (means, weights, covariances)."""
    _base = GMM_CPU_Base

    def compute(self, data):
        self._clf.fit(data)
        return self._clf.means_, self._clf.weights_, self._clf.covariances_
