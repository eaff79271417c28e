"""CPU oracle for the KMeans initialiser of the GMMReg flavour.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package may import this
module: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker.

The reference seeds its flat EM with a third-party estimator that is not part
of ``/root/reference``: ``sklearn.cluster.KMeans(n_clusters=k, random_state=1,
max_iter=50, n_init=1).fit(X).cluster_centers_``
(``src/python/gmmreg_gpu/gmm_impl.py:18-24``; X = the caller's float64 points,
``gmm.py:79``).  The reference does not pin a scikit-learn version (its README
only lists the package), so the algorithm restated here is the one of the
scikit-learn installed in this image, **1.7.2** (``sklearn/cluster/_kmeans.py``:
``KMeans.fit``, ``_kmeans_plusplus``, ``_kmeans_single_lloyd``, and the Cython
helpers ``_relocate_empty_clusters_dense``, ``_average_centers``,
``_center_shift`` of ``_k_means_common.pyx``), written in plain NumPy float64:

* the data are centred by their mean before anything else and the mean is added
  back to the centres at the end;
* ``tol`` is scaled by the mean per-axis variance of X;
* k-means++: the first centre is ``rs.choice(n, p=uniform)``; each later centre
  draws ``2 + int(log k)`` uniforms, turns them into candidate points by
  ``searchsorted(cumsum(closest_dist_sq), u * potential)`` and keeps the
  candidate with the lowest resulting potential (first minimum);
* Lloyd: assignment to the first nearest centre, centres = sum * (1 / count),
  empty clusters take the points farthest from their centres, stop on unchanged
  labels or when the summed squared centre shift is <= tol; without strict
  convergence one more assignment makes labels match the final centres.

Distances are evaluated directly as ``sum((x - c)^2)`` in float64 (scikit-learn
uses the expanded GEMM form; the two differ by ~1e-16 relative, which only
matters for exact ties).

Pinning: ``tests/golden/kmeans_*.npz`` hold scikit-learn's own outputs
(``tools/gen_golden.py``) and ``tests/test_oracle_golden.py`` checks this module
against them (chosen point indices, labels, iteration counts identical, centres
to 1e-12), so parity is pinned to the estimator the reference calls.
"""
from __future__ import annotations

import numpy as np


def _sq_dists(X, C, chunk=4096):
    """[n, k] squared distances, chunked over points."""
    out = np.empty((len(X), len(C)))
    for a in range(0, len(X), chunk):
        d = X[a:a + chunk, None, :] - C[None, :, :]
        out[a:a + chunk] = np.einsum("nkd,nkd->nk", d, d)
    return out


def n_local_trials(k):
    return 2 + int(np.log(k))


def kmeans_plusplus(X, k, rs):
    """sklearn _kmeans_plusplus (dense, unit sample weights).  X must already be centred.
    Returns (centres[k,3], indices[k])."""
    X = np.asarray(X, dtype=np.float64)
    n = len(X)
    trials = n_local_trials(k)
    w = np.ones(n)
    first = rs.choice(n, p=w / w.sum())
    idx = np.full(k, -1, dtype=np.int64)
    idx[0] = first
    closest = ((X - X[first]) ** 2).sum(axis=1)
    pot = closest.sum()
    for c in range(1, k):
        r = rs.uniform(size=trials) * pot
        cand = np.searchsorted(np.cumsum(closest, dtype=np.float64), r)
        np.clip(cand, None, n - 1, out=cand)
        d = np.minimum(closest[None, :], _sq_dists(X[cand], X))       # [trials, n]
        pots = d.sum(axis=1)
        best = int(np.argmin(pots))
        pot = pots[best]
        closest = d[best]
        idx[c] = cand[best]
    return X[idx].copy(), idx


def assign(X, C):
    """First nearest centre per point + squared distance to it."""
    d = _sq_dists(X, C)
    lab = np.argmin(d, axis=1).astype(np.int32)
    return lab, d[np.arange(len(X)), lab]


def _relocate_empty(X, centres_old, sums, counts, labels):
    """_relocate_empty_clusters_dense: every empty cluster takes one of the points that lie
    farthest from their own (old) centre; labels are NOT updated."""
    empty = np.where(counts == 0)[0]
    if len(empty) == 0:
        return
    dist = ((X - centres_old[labels]) ** 2).sum(axis=1)
    far = np.argpartition(dist, -len(empty))[:-len(empty) - 1:-1]
    for new_id, far_idx in zip(empty, far):
        old_id = labels[far_idx]
        sums[old_id] -= X[far_idx]
        sums[new_id] = X[far_idx]
        counts[new_id] = 1.0
        counts[old_id] -= 1.0


def lloyd(X, centres_init, max_iter, tol_abs):
    """_kmeans_single_lloyd.  Returns (labels, inertia, centres, n_iter)."""
    X = np.asarray(X, dtype=np.float64)
    centres = np.array(centres_init, dtype=np.float64)
    k = len(centres)
    labels_old = np.full(len(X), -1, dtype=np.int32)
    strict = False
    n_iter = 0
    for i in range(max_iter):
        labels, _ = assign(X, centres)
        sums = np.zeros((k, 3))
        np.add.at(sums, labels, X)
        counts = np.bincount(labels, minlength=k).astype(np.float64)
        _relocate_empty(X, centres, sums, counts, labels)
        new = sums.copy()
        pos = counts > 0
        new[pos] *= (1.0 / counts[pos])[:, None]
        shift = np.sqrt(((new - centres) ** 2).sum(axis=1))
        centres = new
        n_iter = i + 1
        if np.array_equal(labels, labels_old):
            strict = True
            break
        if (shift ** 2).sum() <= tol_abs:
            break
        labels_old = labels
    if not strict:
        labels, _ = assign(X, centres)
    inertia = ((X - centres[labels]) ** 2).sum()
    return labels, float(inertia), centres, n_iter


def fit(X, k, random_state=1, max_iter=50, tol=1e-4):
    """KMeans(n_clusters=k, random_state=random_state, max_iter=max_iter, n_init=1, tol=tol).fit(X)
    -> dict(cluster_centers_, labels_, inertia_, n_iter_, init_indices)."""
    X = np.asarray(X, dtype=np.float64)
    tol_abs = 0.0 if tol == 0 else float(np.mean(np.var(X, axis=0)) * tol)
    mean = X.mean(axis=0)
    Xc = X - mean
    rs = random_state if isinstance(random_state, np.random.RandomState) else np.random.RandomState(random_state)
    init, idx = kmeans_plusplus(Xc, k, rs)
    labels, inertia, centres, n_iter = lloyd(Xc, init, max_iter, tol_abs)
    return {"cluster_centers_": centres + mean, "labels_": labels, "inertia_": inertia, "n_iter_": n_iter,
            "init_indices": idx}
