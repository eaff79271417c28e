"""Transformations + Gauss transform of the L2 GMMReg path (reference:
src/python/gmmreg_gpu/transforms.py).  The Gauss transform has a vectorised NumPy form (the reference
loops over target rows with ``np.apply_along_axis``) and a device form (``hgmm_gauss_transform``) that
the registration classes use."""
import abc

import numpy as np


class Transformation(abc.ABC):
    def transform(self, points, array_type=None):
        if array_type is not None and isinstance(points, array_type):
            return array_type(self._transform(np.asarray(points)))
        return self._transform(points)

    @abc.abstractmethod
    def _transform(self, points):
        return points


class RigidTransformation(Transformation):
    """scale * X R^T + t   (reference transforms.py:21-40)."""

    def __init__(self, rot=np.identity(3), t=np.zeros(3), scale=1.0):
        self.rot = rot
        self.t = t
        self.scale = scale

    def _transform(self, points):
        return self.scale * np.dot(points, self.rot.T) + self.t

    def inverse(self):
        return RigidTransformation(self.rot.T, -np.dot(self.rot.T, self.t), 1.0 / self.scale)


def _gauss_kernel(source, target, h):
    """e[i, j] = exp(-|target_i - source_j|^2 / h^2), differences taken directly (no expanded form)."""
    d2 = np.zeros((len(target), len(source)))
    for a in range(source.shape[1]):
        diff = target[:, a, None] - source[None, :, a]
        d2 += diff * diff
    d2 /= -(h * h)
    return np.exp(d2, out=d2)


def _gauss_transform_direct(source, target, weights, h):
    """out[i] = sum_j weights[j] exp(-|target_i - source_j|^2 / h^2)   (reference transforms.py:43-49).
    2-D weights [n_w, J]: all rows share ONE kernel matrix -> [n_w, n_target].  The contractions are
    einsum (a threaded BLAS gemv on a J x J matrix costs more than the exponentials)."""
    e = _gauss_kernel(source, target, h)
    if weights.ndim == 1:
        return np.einsum('ij,j->i', e, weights)
    return np.einsum('ij,kj->ki', e, weights)


class Direct(object):
    """Direct evaluation on the host (NumPy)."""

    def __init__(self, source, h):
        self._source = source
        self._h = h

    def compute(self, target, weights):
        return _gauss_transform_direct(self._source, target, weights, self._h)


class DeviceDirect(object):
    """Direct evaluation on the GPU (``hgmm_gauss_transform``, csrc/gmmreg_kernels.hip)."""

    def __init__(self, source, h, ctx):
        self._source = np.ascontiguousarray(source, dtype=np.float64)
        self._h = h
        self._ctx = ctx

    def compute(self, target, weights):
        return self._ctx.gauss_transform(self._source, target, weights, self._h)


class GaussTransform(object):
    """reference transforms.py:60-86 (direct evaluation only, like the reference).  With ``ctx``
    (an ``hgmm_amd.Context``) the sums run on the device, otherwise in NumPy on the host."""

    def __init__(self, source, h, eps=1.0e-4, sw_h=0.01, ctx=None):
        self._m = source.shape[0]
        self._impl = Direct(source, h) if ctx is None else DeviceDirect(source, h, ctx)

    def compute(self, target, weights=None):
        if weights is None:
            weights = np.ones(self._m)
        if weights.ndim == 1:
            return self._impl.compute(target, weights)
        if weights.ndim == 2:
            return self._impl.compute(target, weights)
        raise ValueError("weights.ndim must be 1 or 2.")
