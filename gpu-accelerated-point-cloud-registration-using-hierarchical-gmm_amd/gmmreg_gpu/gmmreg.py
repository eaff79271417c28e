"""L2-distance GMM registration (GMMReg) -- drop-in for ``src/python/gmmreg_gpu/gmmreg.py``.

Both clouds are summarised by a mixture (fitted on the MI355X engine through ``gmm.GMM_GPU``, or by a
one-class SVM's support vectors for the SVR variant); a 7-parameter rigid transform is then found by BFGS on
the L2 distance between the two mixtures, the kernel width annealed between outer rounds.  The optimiser is
host SciPy as in the reference; the J_s x J_t Gauss transform behind every cost evaluation runs on the
device (``hgmm_gauss_transform``).

Layout of this module: the algorithm is three small functions -- :func:`cloud_scale` (the kernel width the
reference derives from the source covariance, gmmreg.py:48-52), :func:`summarise` (features of one cloud, with
the reference's 1e3 weight scaling, gmmreg.py:71-75, 84-89) and :func:`l2_register` (the annealed BFGS loop,
gmmreg.py:62-121) -- and the classes of the reference's API (``L2DistRegistration``, ``RigidGMMReg``,
``RigidSVR``, ``registration_gmmreg``, ``registration_svr``) are thin state holders over them.
"""
import time

import numpy as np
from scipy.optimize import minimize

from . import cost_functions as cf
from . import gmm as ft

WEIGHT_SCALE = 1e3          # both mixtures' weights are multiplied by this before the optimiser sees them


def _points(x):
    return np.asarray(x.points if hasattr(x, "points") else x)


def cloud_scale(data):
    """(det of the sample covariance)^(1 / 2d): the initial kernel width."""
    data = np.asarray(data)
    count, dim = data.shape
    centred = data - data.mean(axis=0)
    return np.power(np.linalg.det(centred.T @ centred / (count - 1)), 1.0 / (2.0 * dim))


def summarise(feature_gen, cloud):
    """-> (centres[J,3] float64, weights[J] float64 x WEIGHT_SCALE) of one cloud."""
    centres, weights = feature_gen.compute(cloud)
    return np.asarray(centres, np.float64), np.asarray(weights, np.float64) * WEIGHT_SCALE


def bfgs_round(cost_fn, x0, source_mix, target_mix, sigma, maxiter, tol, on_step=None):
    """One BFGS solve on fixed mixtures at kernel width ``sigma``."""
    return minimize(cost_fn, x0, args=(*source_mix, *target_mix, sigma), method='BFGS', jac=True, tol=tol,
                    options={'maxiter': maxiter}, callback=on_step)


def l2_register(state, target, maxiter, tol, opt_maxiter, opt_tol):
    """The outer loop: re-summarise the source, solve, anneal, until the cost stops moving.
    ``state`` is the registration object (it owns sigma, the feature generator and the callbacks)."""
    state._feature_gen.init()
    target_mix = summarise(state._feature_gen, target)
    x = state._cost_fn.initial()
    previous = None
    for _ in range(maxiter):
        source_mix = summarise(state._feature_gen, state._source)
        res = bfgs_round(state._cost_fn, x, source_mix, target_mix, state._sigma, opt_maxiter, opt_tol,
                         state.optimization_cb)
        state._annealing()
        state._feature_gen.annealing()
        x = res.x
        if previous is not None and abs(res.fun - previous) < tol:
            break
        previous = res.fun
    return state._cost_fn.to_transformation(x)


class L2DistRegistration(object):
    """reference gmmreg.py:14-121 (same constructor, attributes and methods)."""

    def __init__(self, source, feature_gen, cost_fn, sigma=1.0, delta=0.9, use_estimated_sigma=True,
                 verbose=False):
        self._feature_gen, self._cost_fn = feature_gen, cost_fn
        self._sigma, self._delta = sigma, delta
        self._use_estimated_sigma = use_estimated_sigma
        self._callbacks = []
        self._verbose = verbose
        self._source = None
        if source is not None:
            self.set_source(source)

    def set_source(self, source):
        self._source = source
        if self._use_estimated_sigma:
            self._estimate_sigma(source)

    def set_callbacks(self, callbacks):
        self._callbacks.extend(callbacks)

    def _estimate_sigma(self, data):
        self._sigma = cloud_scale(data)
        if self._verbose:
            print("Estimated Sigma: ", self._sigma)

    def _annealing(self):
        self._sigma *= self._delta

    def optimization_cb(self, x):
        if self._callbacks:
            tf_result = self._cost_fn.to_transformation(x)
            for c in self._callbacks:
                c(tf_result)

    def optimise(self, mu_source, phi_source, mu_target, phi_target, x_ini, opt_maxiter=10, opt_tol=1.0e-5):
        """One BFGS solve on given mixtures (the inner step of :meth:`registration`)."""
        return bfgs_round(self._cost_fn, x_ini, (mu_source, phi_source), (mu_target, phi_target), self._sigma,
                          opt_maxiter, opt_tol, self.optimization_cb)

    def registration(self, target, maxiter=1, tol=1.0e-3, opt_maxiter=10, opt_tol=1.0e-5):
        t0 = time.time()
        out = l2_register(self, target, maxiter, tol, opt_maxiter, opt_tol)
        if self._verbose:
            print("Overall Time taken: ", time.time() - t0)
        return out


def _default_cost(ctx):
    from .._native import default_context
    return cf.RigidCostFunction(ctx=ctx or default_context())


class RigidGMMReg(L2DistRegistration):
    """reference gmmreg.py:138-147: GMM_GPU features (10 EM iterations), at most 0.8 N components."""

    def __init__(self, source, sigma=1.0, delta=0.9, n_gmm_components=50, use_estimated_sigma=True,
                 verbose=False, ctx=None):
        k = min(n_gmm_components, int(source.shape[0] * 0.8))
        super().__init__(source, ft.GMM_GPU(k, max_iter=10), _default_cost(ctx), sigma, delta, use_estimated_sigma,
                         verbose)


class RigidSVR(L2DistRegistration):
    """reference gmmreg.py:123-136: the clouds are summarised by the support vectors of a one-class SVM
    (third-party estimator, scikit-learn on the host) instead of a GMM; cost, gradient and Gauss transform are
    the same device path as RigidGMMReg.  The SVM's RBF width follows the kernel width."""

    def __init__(self, source, sigma=1.0, delta=0.9, gamma=0.5, nu=0.1, use_estimated_sigma=True,
                 verbose=False, ctx=None):
        super().__init__(source, ft.OneClassSVM(source.shape[1], sigma, gamma, nu), _default_cost(ctx), sigma, delta,
                         use_estimated_sigma, verbose)

    def _estimate_sigma(self, data):
        super()._estimate_sigma(data)
        self._feature_gen._sigma = self._sigma
        self._feature_gen._gamma = 1.0 / (2.0 * np.square(self._sigma))


def _run(kind, source, target, tf_type_name, callbacks, reg_args, kargs):
    if tf_type_name != 'rigid':
        raise ValueError('Unknown transform type %s' % tf_type_name)
    reg = kind(_points(source), **kargs)
    reg.set_callbacks(callbacks)
    return reg.registration(_points(target), *reg_args)


def registration_gmmreg(source, target, tf_type_name='rigid', callbacks=[], **kargs):
    """reference gmmreg.py:149-157: returns the RigidTransformation mapping source onto target."""
    return _run(RigidGMMReg, source, target, tf_type_name, callbacks, (), kargs)


def registration_svr(source, target, tf_type_name='rigid', maxiter=1, tol=1.0e-3, opt_maxiter=50,
                     opt_tol=1.0e-3, callbacks=[], **kargs):
    """reference gmmreg.py:159-169."""
    return _run(RigidSVR, source, target, tf_type_name, callbacks, (maxiter, tol, opt_maxiter, opt_tol), kargs)
