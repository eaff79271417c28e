"""Rotation helpers of the L2 GMMReg path (reference: src/python/gmmreg_gpu/so.py and the
third-party ``transformations.quaternion_matrix`` it calls, which this image lacks).

Quaternions are (w, x, y, z).  With n = |q|^2 the rotation is  R = I + (2/n) A(q),
  A = [[-(yy+zz), xy-wz, xz+wy], [xy+wz, -(xx+zz), yz-wx], [xz-wy, yz+wx, -(xx+yy)]].
"""
import numpy as np


def quaternion_matrix(q):
    """4x4 homogeneous rotation of a (not necessarily unit) quaternion; identity for |q| ~ 0."""
    q = np.array(q, dtype=np.float64, copy=True)
    n = np.dot(q, q)
    m = np.identity(4)
    if n < np.finfo(float).eps * 4.0:
        return m
    m[:3, :3] = np.identity(3) + (2.0 / n) * _a_matrix(q)
    return m


def _a_matrix(q):
    w, x, y, z = q
    return np.array([[-(y * y + z * z), x * y - w * z, x * z + w * y],
                     [x * y + w * z, -(x * x + z * z), y * z - w * x],
                     [x * z - w * y, y * z + w * x, -(x * x + y * y)]])


def _da_matrices(q):
    """dA/dq_k for k = w, x, y, z  -> [4,3,3]."""
    w, x, y, z = q
    return np.array([
        [[0, -z, y], [z, 0, -x], [-y, x, 0]],
        [[0, y, z], [y, -2 * x, -w], [z, w, -2 * x]],
        [[-2 * y, x, w], [x, 0, z], [-w, z, -2 * y]],
        [[-2 * z, -w, x], [w, -2 * z, y], [x, y, 0]],
    ], dtype=np.float64)


def diff_rot_from_quaternion(q, reference_quirks=True):
    """dR(q)/dq = [dR/dw, dR/dx, dR/dy, dR/dz]  (reference so.py:4-59).

    Diagonal entries are the exact derivative (2/n) dA_ii/dq_k - (4 q_k / n^2) A_ii.  For the
    off-diagonal entries the reference evaluates  (2/n) dA_ij/dq_k - 2 q_k R_ij / n^2, which
    equals the exact derivative only for unit quaternions (it is short of a factor n in the second
    term), and two of its diagonal entries, dR22/dy and dR22/dz, carry swapped squared terms
    (so.py:23-24: -4y(xx+yy)/n^2 and 4z(zz+ww)/n^2 instead of -4y(ww+zz)/n^2 and 4z(xx+yy)/n^2).
    Those behaviours are reproduced (``reference_quirks=True``) so that BFGS trajectories match
    the reference's; ``reference_quirks=False`` gives the exact Jacobian."""
    q = np.asarray(q, dtype=np.float64)
    n = float(np.dot(q, q))
    a = _a_matrix(q)
    da = _da_matrices(q)
    rot = np.identity(3) + (2.0 / n) * a
    exact = (2.0 / n) * da - (4.0 / (n * n)) * q[:, None, None] * a[None]
    quirk = (2.0 / n) * da - (2.0 / (n * n)) * q[:, None, None] * rot[None]
    if not reference_quirks:
        return exact
    eye = np.identity(3, dtype=bool)[None]
    d = np.where(eye, exact, quirk)
    w, x, y, z = q
    d[2, 2, 2] = -4.0 * y * (x * x + y * y) / (n * n)
    d[3, 2, 2] = 4.0 * z * (z * z + w * w) / (n * n)
    return d
