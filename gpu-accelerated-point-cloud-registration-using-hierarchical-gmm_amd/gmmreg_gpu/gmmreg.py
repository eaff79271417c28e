"""L2-distance GMM registration (GMMReg) -- drop-in for the reference's
``src/python/gmmreg_gpu/gmmreg.py``: both clouds are summarised by a GMM (fitted on the MI355X
engine through ``gmm.GMM_GPU``), then a 7-parameter rigid transform is found by BFGS on the L2
distance between the two mixtures (host SciPy as in the reference; the J_s x J_t Gauss transform
behind every cost evaluation runs on the device)."""
import time

import numpy as np
from scipy.optimize import minimize

from . import cost_functions as cf
from . import gmm as ft


def _points(x):
    return np.asarray(x.points if hasattr(x, "points") else x)


class L2DistRegistration(object):
    """reference gmmreg.py:14-121."""

    def __init__(self, source, feature_gen, cost_fn, sigma=1.0, delta=0.9, use_estimated_sigma=True,
                 verbose=False):
        self._source = source
        self._feature_gen = feature_gen
        self._cost_fn = cost_fn
        self._sigma = sigma
        self._delta = delta
        self._use_estimated_sigma = use_estimated_sigma
        self._callbacks = []
        self._verbose = verbose
        if self._source is not None and self._use_estimated_sigma:
            self._estimate_sigma(self._source)

    def set_source(self, source):
        self._source = source
        if self._use_estimated_sigma:
            self._estimate_sigma(self._source)

    def set_callbacks(self, callbacks):
        self._callbacks.extend(callbacks)

    def _estimate_sigma(self, data):
        ndata, ndim = data.shape
        data_hat = data - np.mean(data, axis=0)
        self._sigma = np.power(np.linalg.det(np.dot(data_hat.T, data_hat) / (ndata - 1)), 1.0 / (2.0 * ndim))
        if self._verbose:
            print("Estimated Sigma: ", self._sigma)

    def _annealing(self):
        self._sigma *= self._delta

    def optimization_cb(self, x):
        tf_result = self._cost_fn.to_transformation(x)
        for c in self._callbacks:
            c(tf_result)

    def optimise(self, mu_source, phi_source, mu_target, phi_target, x_ini, opt_maxiter=10, opt_tol=1.0e-5):
        """One BFGS solve on fixed mixtures (the inner step of :meth:`registration`)."""
        args = (mu_source, phi_source, mu_target, phi_target, self._sigma)
        return minimize(self._cost_fn, x_ini, args=args, method='BFGS', jac=True, tol=opt_tol,
                        options={'maxiter': opt_maxiter}, callback=self.optimization_cb)

    def registration(self, target, maxiter=1, tol=1.0e-3, opt_maxiter=10, opt_tol=1.0e-5):
        start = time.time()
        f = None
        x_ini = self._cost_fn.initial()
        self._feature_gen.init()
        mu_target, phi_target = self._feature_gen.compute(target)
        phi_target = np.asarray(phi_target) * 1e3                 # reference gmmreg.py:75
        res = None
        for _ in range(maxiter):
            mu_source, phi_source = self._feature_gen.compute(self._source)
            phi_source = np.asarray(phi_source) * 1e3             # reference gmmreg.py:89
            res = self.optimise(np.asarray(mu_source, np.float64), np.asarray(phi_source, np.float64),
                                np.asarray(mu_target, np.float64), np.asarray(phi_target, np.float64),
                                x_ini, opt_maxiter, opt_tol)
            self._annealing()
            self._feature_gen.annealing()
            if f is not None and abs(res.fun - f) < tol:
                break
            f = res.fun
            x_ini = res.x
        if self._verbose:
            print("Overall Time taken: ", time.time() - start)
        return self._cost_fn.to_transformation(res.x)


class RigidGMMReg(L2DistRegistration):
    """reference gmmreg.py:138-147 (GMM_GPU feature, 10 EM iterations)."""

    def __init__(self, source, sigma=1.0, delta=0.9, n_gmm_components=50, use_estimated_sigma=True,
                 verbose=False, ctx=None):
        from .._native import default_context
        n_gmm_components = min(n_gmm_components, int(source.shape[0] * 0.8))
        super(RigidGMMReg, self).__init__(source, ft.GMM_GPU(n_gmm_components, max_iter=10),
                                          cf.RigidCostFunction(ctx=ctx or default_context()), sigma, delta,
                                          use_estimated_sigma, verbose)


def registration_gmmreg(source, target, tf_type_name='rigid', callbacks=[], **kargs):
    """reference gmmreg.py:149-157: returns the RigidTransformation mapping source onto target."""
    if tf_type_name != 'rigid':
        raise ValueError('Unknown transform type %s' % tf_type_name)
    gmmreg = RigidGMMReg(_points(source), **kargs)
    gmmreg.set_callbacks(callbacks)
    return gmmreg.registration(_points(target))


class RigidSVR(L2DistRegistration):
    """reference gmmreg.py:123-136: support-vector registration -- the two clouds are summarised by the
    support vectors of a one-class SVM (third-party estimator, scikit-learn on the host) instead of a GMM;
    the L2 cost, its gradient and the Gauss transform behind it are the same device path as RigidGMMReg."""

    def __init__(self, source, sigma=1.0, delta=0.9, gamma=0.5, nu=0.1, use_estimated_sigma=True,
                 verbose=False, ctx=None):
        from .._native import default_context
        super(RigidSVR, self).__init__(source, ft.OneClassSVM(source.shape[1], sigma, gamma, nu),
                                       cf.RigidCostFunction(ctx=ctx or default_context()), sigma, delta,
                                       use_estimated_sigma, verbose)

    def _estimate_sigma(self, data):
        super(RigidSVR, self)._estimate_sigma(data)
        self._feature_gen._sigma = self._sigma
        self._feature_gen._gamma = 1.0 / (2.0 * np.square(self._sigma))


def registration_svr(source, target, tf_type_name='rigid', maxiter=1, tol=1.0e-3, opt_maxiter=50,
                     opt_tol=1.0e-3, callbacks=[], **kargs):
    """reference gmmreg.py:159-169."""
    if tf_type_name != 'rigid':
        raise ValueError('Unknown transform type %s' % tf_type_name)
    svr = RigidSVR(_points(source), **kargs)
    svr.set_callbacks(callbacks)
    return svr.registration(_points(target), maxiter, tol, opt_maxiter, opt_tol)
