"""Rigid transformation + Gauss transform of the L2 GMMReg path (drop-in for
``src/python/gmmreg_gpu/transforms.py``: ``RigidTransformation``, ``GaussTransform``).

The Gauss transform  out[i] = sum_j w[j] exp(-|t_i - s_j|^2 / h^2)  is the only O(J_s J_t) piece of the cost.
Two evaluators share one interface (``GaussTransform(source, h, ctx=...).compute(target, weights)``):
``kernel_sums_host`` (vectorised NumPy; the reference loops over target rows with ``np.apply_along_axis``,
transforms.py:43-49) and the device kernel ``hgmm_gauss_transform`` (csrc/gmmreg_kernels.hip) used by the
registration classes.  Several weight rows [n_w, J] share one kernel matrix.
"""
import numpy as np


class Transformation(object):
    """Base of the transformation objects callbacks receive: ``transform(points)`` applies ``_transform``;
    ``array_type`` re-wraps the result for callers that pass an Open3D vector (transforms.py:12-16)."""

    def _transform(self, points):
        raise NotImplementedError

    def transform(self, points, array_type=None):
        wrap = array_type is not None and isinstance(points, array_type)
        out = self._transform(np.asarray(points) if wrap else points)
        return array_type(out) if wrap else out


class RigidTransformation(Transformation):
    """x -> scale * R x + t on row vectors (transforms.py:21-40)."""

    def __init__(self, rot=np.identity(3), t=np.zeros(3), scale=1.0):
        self.rot, self.t, self.scale = rot, t, scale

    def _transform(self, points):
        return self.scale * (points @ self.rot.T) + self.t

    def inverse(self):
        back = self.rot.T
        return RigidTransformation(back, -(back @ self.t), 1.0 / self.scale)


def kernel_matrix(source, target, h):
    """e[i, j] = exp(-|target_i - source_j|^2 / h^2), differences taken axis by axis (no expanded form)."""
    d2 = np.zeros((len(target), len(source)))
    for axis in range(source.shape[1]):
        delta = target[:, axis, None] - source[None, :, axis]
        d2 += delta * delta
    return np.exp(d2 / -(h * h), out=d2)


def kernel_sums_host(source, target, weights, h):
    """Gauss transform on the host.  einsum, not a threaded BLAS gemv: on a J x J matrix the thread start-up costs
    more than the exponentials."""
    e = kernel_matrix(source, target, h)
    return np.einsum('ij,j->i', e, weights) if weights.ndim == 1 else np.einsum('ij,kj->ki', e, weights)


class GaussTransform(object):
    """reference transforms.py:60-86 (direct evaluation only, like the reference; ``eps`` / ``sw_h`` are accepted
    and ignored as there).  ``ctx`` = an ``hgmm_amd.Context``: evaluate on that device."""

    def __init__(self, source, h, eps=1.0e-4, sw_h=0.01, ctx=None):
        self._source = np.ascontiguousarray(source, dtype=np.float64)
        self._h = h
        self._ctx = ctx

    def compute(self, target, weights=None):
        if weights is None:
            weights = np.ones(len(self._source))
        if weights.ndim not in (1, 2):
            raise ValueError("weights.ndim must be 1 or 2.")
        if self._ctx is not None:
            return self._ctx.gauss_transform(self._source, target, weights, self._h)
        return kernel_sums_host(self._source, np.asarray(target, dtype=np.float64), weights, self._h)
