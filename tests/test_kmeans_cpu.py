"""Host-side pieces of the KMeans initialiser that must reproduce NumPy bit for bit (no GPU needed)."""
import numpy as np
import pytest

from hgmm_amd.kmeans import column_mean_var


@pytest.mark.parametrize("n", [1, 2, 7, 1000, 40256, 300001])
def test_column_mean_var_equals_numpy_bit_for_bit(n):
    """scikit-learn's KMeans centres X by ``X.mean(axis=0)`` and scales ``tol`` by ``np.var(X, axis=0)``
    (sklearn/cluster/_kmeans.py, ``_tolerance`` / ``fit``); the faster evaluation order used by the mirror has to
    give the very same doubles, or seeds and stop iteration could differ from scikit-learn's."""
    rs = np.random.RandomState(n)
    X = rs.rand(n, 3) * rs.uniform(0.01, 80.0) + rs.uniform(-30, 30, size=3)
    mean, var, Xc = column_mean_var(X)
    assert np.array_equal(mean, X.mean(axis=0))
    assert np.array_equal(var, np.var(X, axis=0))
    assert np.array_equal(Xc, X - X.mean(axis=0))
    assert Xc.flags["C_CONTIGUOUS"] and Xc.shape == (n, 3)


def test_column_mean_var_accepts_fortran_and_strided_views():
    rs = np.random.RandomState(3)
    base = rs.randn(5000, 6)
    for X in (np.asfortranarray(base[:, :3]), base[:, ::2], base[::2, 3:]):
        mean, var, Xc = column_mean_var(X)
        # NumPy's own order of summation depends on the memory layout; the C-ordered copy is the reference case
        Xd = np.ascontiguousarray(X)
        assert np.array_equal(mean, Xd.mean(axis=0)) and np.array_equal(var, np.var(Xd, axis=0))
