"""L2 distance between two Gaussian mixtures and its rigid (quaternion + translation) cost
(drop-in for ``src/python/gmmreg_gpu/cost_functions.py``: ``compute_l2_dist``, ``RigidCostFunction``).

With both mixtures isotropic of width sigma, the cross term of the L2 distance is a Gauss transform of width
sqrt(2) sigma: f = -sum_s phi_s G(mu_s), G(x) = sum_t (phi_t / z) exp(-|x - mu_t|^2 / 2 sigma^2); its gradient
with respect to the source centres needs the same kernel against the weight rows phi_t mu_t / z.  Both come out
of ONE kernel matrix here (the reference evaluates it 1 + 3 times, cost_functions.py:29-40); with a device
context it is `hgmm_gauss_transform`.  The 7-parameter chain rule stays in NumPy.
"""
import numpy as np

from . import so
from . import transforms as tf


def compute_l2_dist(mu_source, phi_source, mu_target, phi_target, sigma, ctx=None):
    """-> (f, df/dmu_source [J_s, 3]).  ``ctx``: evaluate the Gauss transform on that device context."""
    dim = mu_source.shape[1]
    z = np.power(2.0 * np.pi * sigma ** 2, dim * 0.5)
    rows = np.vstack([phi_target, phi_target * mu_target.T]) / z           # [1 + dim, J_t]
    sums = tf.GaussTransform(mu_target, np.sqrt(2.0) * sigma, ctx=ctx).compute(mu_source, rows)
    g0, g1 = sums[0], sums[1:]                                             # sum_t w e  and  sum_t w mu_t e
    grad = (phi_source * (g0 * mu_source.T - g1)).T / (2.0 * sigma ** 2)
    return -(phi_source @ g0), grad


class CostFunction(object):
    """Interface BFGS drives: ``initial()`` -> theta0, ``__call__(theta, *mixtures)`` -> (f, grad),
    ``to_transformation(theta)`` (cost_functions.py:11-26)."""

    def __init__(self, tf_type):
        self._tf_type = tf_type

    def to_transformation(self, theta):
        raise NotImplementedError

    def initial(self):
        raise NotImplementedError

    def __call__(self, theta, *args):
        raise NotImplementedError


class RigidCostFunction(CostFunction):
    """theta = (qw, qx, qy, qz, tx, ty, tz)   (cost_functions.py:43-68)."""

    def __init__(self, ctx=None):
        super().__init__(tf.RigidTransformation)
        self._ctx = ctx                     # device context for the Gauss transform (None: host NumPy)

    def to_transformation(self, theta):
        return self._tf_type(so.quaternion_matrix(theta[:4])[:3, :3], theta[4:7])

    def initial(self):
        return np.array([1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])

    def __call__(self, theta, *args):
        mu_source, phi_source, mu_target, phi_target, sigma = args
        moved = self.to_transformation(theta).transform(mu_source)
        f, g = compute_l2_dist(moved, phi_source, mu_target, phi_target, sigma, ctx=self._ctx)
        # d f / d q_k = sum_{a,b} (g^T mu_source)[a, b] dR[a, b] / d q_k ;  d f / d t = column sums of g
        d_quat = np.einsum('ab,kab->k', g.T @ mu_source, so.diff_rot_from_quaternion(theta[:4]))
        return f, np.concatenate([d_quat, g.sum(axis=0)])
