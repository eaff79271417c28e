"""L2 distance between two Gaussian mixtures + its rigid (quaternion, translation) cost
(reference: src/python/gmmreg_gpu/cost_functions.py).  The J_s x J_t Gauss transform runs on the
device when the cost function holds a context, the 7-parameter chain rule stays in NumPy."""
import abc

import numpy as np

from . import so
from . import transforms as tf


class CostFunction(abc.ABC):
    def __init__(self, tf_type):
        self._tf_type = tf_type

    @abc.abstractmethod
    def to_transformation(self, theta):
        return None

    @abc.abstractmethod
    def initial(self):
        return None

    @abc.abstractmethod
    def __call__(self, theta, *args):
        return None, None


def compute_l2_dist(mu_source, phi_source, mu_target, phi_target, sigma, ctx=None):
    """-> (-phi_s . G(mu_s), gradient wrt mu_s [J_s,3])   (reference cost_functions.py:29-40).
    ``ctx``: evaluate the Gauss transform on that device context instead of in NumPy."""
    z = np.power(2.0 * np.pi * sigma ** 2, mu_source.shape[1] * 0.5)
    gtrans = tf.GaussTransform(mu_target, np.sqrt(2.0) * sigma, ctx=ctx)
    # one kernel matrix for both transforms (the reference evaluates it 1 + 3 times)
    both = gtrans.compute(mu_source, np.vstack([phi_target / z, phi_target * mu_target.T / z]))
    phi_j_e, phi_mu_j_e = both[0], both[1:].T
    g = (phi_source * phi_j_e * mu_source.T - phi_source * phi_mu_j_e.T).T / (2.0 * sigma ** 2)
    return -np.dot(phi_source, phi_j_e), g


class RigidCostFunction(CostFunction):
    """theta = (qw, qx, qy, qz, tx, ty, tz)   (reference cost_functions.py:43-68)."""

    def __init__(self, ctx=None):
        self._tf_type = tf.RigidTransformation
        self._ctx = ctx                     # device context for the Gauss transform (None: host NumPy)

    def to_transformation(self, theta):
        rot = so.quaternion_matrix(theta[:4])[:3, :3]
        return self._tf_type(rot, theta[4:7])

    def initial(self):
        x0 = np.zeros(7)
        x0[0] = 1.0
        return x0

    def __call__(self, theta, *args):
        mu_source, phi_source, mu_target, phi_target, sigma = args
        tf_obj = self.to_transformation(theta)
        t_mu_source = tf_obj.transform(mu_source)
        f, g = compute_l2_dist(t_mu_source, phi_source, mu_target, phi_target, sigma, ctx=self._ctx)
        d_rot = so.diff_rot_from_quaternion(theta[:4])
        gtm0 = np.dot(g.T, mu_source)
        grad = np.concatenate([(gtm0 * d_rot).sum(axis=(1, 2)), g.sum(axis=0)])
        return f, grad
