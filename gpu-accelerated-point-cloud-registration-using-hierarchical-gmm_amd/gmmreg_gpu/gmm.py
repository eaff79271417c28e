"""Drop-in for the reference's ``src/python/gmmreg_gpu/gmm.py`` (flavour "G"): ``compute``
returns ``(means, weights)`` only (reference gmm.py:57,70); covariances start at 0.1
(gmm.py:80); host NumPy results (gmm.py:95)."""
import numpy as np

from ..gmm_waymo.gmm import Feature, GMM_Sklearn, _cpu_name_notice  # noqa: F401
from ..gmm_waymo.gmm import OneClassSVM as _SklearnOneClassSVM
from . import gmm_impl
from .gmm_impl import train_gmm, init_gmm_params, timer, predict, asarray  # noqa: F401


class GMM_GPU_Base:
    _label = 'GPU GMM TRAIN'

    def __init__(self, num_components, max_iter=30, tol=1e-4):
        self.num_components = num_components
        self.max_iter = max_iter
        self.tol = tol

    def fit(self, X, init=None):
        X = np.asarray(X)
        means, weights = init if init is not None else init_gmm_params(X, self.num_components)
        covs = 0.1 * np.ones((self.num_components, 3), dtype=np.float32)
        dev_X = asarray(X.astype(np.float32))
        with timer(self._label):
            inv, mu, w, cov, lls = train_gmm(dev_X, self.max_iter, self.tol, np.asarray(means, np.float32),
                                             covs, np.asarray(weights, np.float32))
        self.means_, self.covariances_, self.weights_ = mu, cov, w
        self.lls, self.inv_covs = lls, inv
        if len(lls):
            print("\nLog Likelihood Min-Max:\n\n", np.min(lls), np.max(lls))
        return self

    def predict(self, X):
        return predict(np.asarray(X).astype(np.float32), self.inv_covs, self.means_, self.weights_)


class GMM_CPU_Base(GMM_GPU_Base):
    """The reference's GMM_CPU_Base.fit is broken (unpacks 3 values from a 2-tuple,
    gmmreg_gpu/gmm.py:117); this one works and, like everything here, runs on the GPU engine."""
    _label = 'CPU GMM TRAIN'

    def __init__(self, *args, **kwargs):
        _cpu_name_notice("GMM_CPU_Base")
        super().__init__(*args, **kwargs)


class GMM_GPU(Feature):
    _base = GMM_GPU_Base

    def __init__(self, n_gmm_components=100, max_iter=30, tol=1e-4):
        self._n_gmm_components = n_gmm_components
        self.max_iter = max_iter
        self.tol = tol

    def init(self):
        self._clf = self._base(self._n_gmm_components, max_iter=self.max_iter, tol=self.tol)

    def compute(self, data):
        self._clf.fit(data)
        return self._clf.means_, self._clf.weights_

    def predict(self, data):
        return self._clf.predict(data)


class GMM_CPU(GMM_GPU):
    _base = GMM_CPU_Base


class OneClassSVM(_SklearnOneClassSVM):
    """reference gmmreg_gpu/gmm.py:141-167.  That file imports the third-party ``thundersvm`` at module
    level (gmm.py:12) but builds the estimator from scikit-learn's ``svm.OneClassSVM`` (gmm.py:158), which
    is what this class does; ``RigidSVR`` / ``registration_svr`` (gmmreg.py:123-136, 159-169) use it."""
