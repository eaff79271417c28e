"""Drop-in for the reference's ``src/python/gmm_waymo/src/gmm.py``: the sklearn-style feature
extractors that callers (run_gmm_static.py:35-49, run_gmm_waymo_gpu.py, gmmreg.py:71,84,144)
use -- ``init()``, ``compute(data)``, ``predict(data)``, ``fit(X)``.

``GMM_GPU`` / ``GMM_GPU_Base`` and ``GMM_CPU`` / ``GMM_CPU_Base`` all run on the MI355X engine: this
package has no CPU compute path.  The ``GMM_CPU`` names are kept for API compatibility (3-tuple
``compute``, reference gmm.py:103-149); they say so once per process (``EngineNotice`` warning) so that a
caller timing "CPU vs GPU" is not comparing the engine with itself unknowingly.  ``GMM_Sklearn`` and
``OneClassSVM`` wrap scikit-learn exactly as the reference does (gmm.py:29-44, 151-177); scikit-learn is
imported when ``init()`` is called.
"""
import abc
import warnings

import numpy as np

from . import gmm_impl
from .gmm_impl import train_gmm, init_gmm_params, timer, predict, asarray, DevicePoints  # noqa: F401


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
        if isinstance(X, DevicePoints):                   # already resident (hgmm_amd.asarray): no upload
            # the reference's initialiser samples from the HOST array (gmm_impl.py:26-41): one download of the
            # float32 rows when the caller brings no initial parameters
            means, weights, covs = init if init is not None else self._init_params(X.get())
            dev_X = X
        else:
            X = np.asarray(X)
            means, weights, covs = init if init is not None else self._init_params(X)
            dev_X = asarray(X.astype(np.float32))
        with timer(self._label, getattr(dev_X, "ctx", None)):
            inv, mu, w, cov, lls = train_gmm(dev_X, self.max_iter, self.tol,
                                             np.asarray(means, np.float32), np.asarray(covs, np.float32),
                                             np.asarray(weights, np.float32), cov_type=self.cov_type)
        self.means_, self.covariances_, self.weights_ = mu, cov, w
        self.lls, self.inv_covs = lls, inv
        if self._verbose and len(lls):
            print("\nLog Likelihood Min-Max:\n\n", np.min(lls), np.max(lls))
        return self

    def predict(self, X):
        """Host array in -> NumPy int64 labels; a resident cloud (DevicePoints) in -> the labels as a DeviceArray,
        nothing downloaded (the reference returns the CuPy array and its caller does ``cupy.asnumpy``,
        run_gmm_waymo_gpu.py:52-55)."""
        if not isinstance(X, DevicePoints):
            X = np.asarray(X).astype(np.float32)
        return predict(X, self.inv_covs, self.means_, self.weights_, cov_type=self.cov_type)


class EngineNotice(UserWarning):
    """Raised (as a warning) when an API name promises a CPU path this package does not have."""


def _cpu_name_notice(name):
    warnings.warn("%s is an API-compatibility name: the fit runs on the MI355X engine (hgmm_amd has no CPU "
                  "compute path); for a CPU run use the reference's own gmm.py" % name, EngineNotice, stacklevel=3)


class GMM_CPU_Base(GMM_GPU_Base):
    """reference gmm.py:120-149 -- same engine as GMM_GPU_Base, quieter (no min/max print)."""
    _label = 'CPU GMM TRAIN'
    _verbose = False

    def __init__(self, *args, **kwargs):
        _cpu_name_notice("GMM_CPU_Base")
        super().__init__(*args, **kwargs)


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


class GMM_Sklearn(Feature):
    """reference gmm.py:29-44: scikit-learn's GaussianMixture (k-means initialisation) behind the
    Feature protocol; ``compute`` -> (means, weights, covariances, None).  Third-party estimator on the
    host, kept so that ``from gmm import GMM_CPU, GMM_Sklearn, GMM_GPU`` (run_gmm_static.py:5) resolves."""

    def __init__(self, n_gmm_components=50, max_iter=30, tol=1e-4, cov_type='diag'):
        self._n_gmm_components = n_gmm_components
        self.max_iter = max_iter
        self.tol = tol
        self.cov_type = cov_type

    def init(self):
        from sklearn import mixture
        self._clf = mixture.GaussianMixture(n_components=self._n_gmm_components, max_iter=self.max_iter,
                                            init_params='kmeans', covariance_type=self.cov_type)

    def compute(self, data):
        self._clf.fit(data)
        return self._clf.means_, self._clf.weights_, self._clf.covariances_, None

    def predict(self, data):
        return self._clf.predict(data)


class OneClassSVM(Feature):
    """reference gmm.py:151-177: support vectors of a one-class SVM as mixture centres, dual
    coefficients x (2 pi sigma^2)^(ndim/2) as weights.  scikit-learn on the host."""

    def __init__(self, ndim, sigma, gamma=0.5, nu=0.05, delta=10.0):
        self._ndim = ndim
        self._sigma = sigma
        self._gamma = gamma
        self._nu = nu
        self._delta = delta

    def init(self):
        from sklearn import svm
        self._clf = svm.OneClassSVM(nu=self._nu, kernel="rbf", gamma=self._gamma)

    def compute(self, data):
        self._clf.fit(data)
        z = np.power(2.0 * np.pi * self._sigma ** 2, self._ndim * 0.5)
        return self._clf.support_vectors_, self._clf.dual_coef_[0] * z

    def annealing(self):
        self._gamma *= self._delta
