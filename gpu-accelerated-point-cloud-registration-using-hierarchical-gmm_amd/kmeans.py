"""KMeans initialiser of the GMMReg flavour on the device.

The reference seeds its flat EM with ``sklearn.cluster.KMeans(n_clusters=k, random_state=1,
max_iter=50, n_init=1).fit(X).cluster_centers_`` (``src/python/gmmreg_gpu/gmm_impl.py:18-24``).
``KMeans`` below takes the same constructor arguments, exposes the attributes scikit-learn's
estimator leaves behind (``cluster_centers_``, ``labels_``, ``inertia_``, ``n_iter_``) and
follows scikit-learn 1.7's ``KMeans.fit`` step by step; the two O(N k) parts -- greedy k-means++
seeding and the Lloyd assignment / per-cluster sums -- run in HIP kernels
(``csrc/kmeans_kernels.hip``) on the float64 cloud, everything that touches only k x 3 numbers
(random draws, centre update, stop rule, empty-cluster relocation) stays here in NumPy:

* X is centred by its mean first, the mean is added back to the centres at the end;
* ``tol`` is scaled by the mean per-axis variance of X;
* the random stream is NumPy's ``RandomState(random_state)`` consumed in scikit-learn's order
  (one ``choice`` for the first centre, then ``2 + int(log k)`` uniforms per centre), so the same
  points are chosen as seeds;
* Lloyd: first nearest centre, centres = sum * (1 / count), empty clusters take the points
  farthest from their centre (``np.argpartition`` order, like scikit-learn), stop on unchanged
  labels or summed squared centre shift <= tol, one more assignment if the stop was not strict.

Arithmetic is float64 whatever the input dtype (scikit-learn keeps float32 input in float32).
With an RCCL communicator attached to the context, X is this rank's shard of one joint clustering:
mean, variance, per-cluster sums, inertia and the changed-label count are all-reduced, the seeds
are drawn on rank 0's shard; empty-cluster relocation stays local to each shard (single-GPU exact).
"""
from __future__ import annotations

import numpy as np

from ._native import Context, default_context


def _random_state(seed):
    if isinstance(seed, np.random.RandomState):
        return seed
    if seed is None:
        return np.random.mtrand._rand
    return np.random.RandomState(seed)


def relocate_empty_sharded(allreduce, rank, labels, dist, Xc, sums, counts):
    """Empty-cluster relocation when the cloud is sharded over ranks: the globally farthest points
    are found with one max-reduction per empty cluster (ties go to the lowest rank), the chosen point
    travels as a sum with zeros elsewhere.  Every rank ends with identical sums / counts.
    ``allreduce(values, op)`` is the communicator's reduction ("sum" / "max")."""
    empty = np.where(counts == 0)[0]
    if len(empty) == 0:
        return
    order = np.argsort(-dist, kind="stable")[:len(empty)]      # local candidates, farthest first
    used = 0
    for new_id in empty:
        d_loc = float(dist[order[used]]) if used < len(order) else -1.0
        d_max = allreduce([d_loc], "max")[0]
        mine = d_loc == d_max
        owner = -allreduce([-float(rank) if mine else -1.0e9], "max")[0]
        payload = np.zeros(4)
        if mine and owner == rank:
            i = order[used]
            used += 1
            payload[:3] = Xc[i]
            payload[3] = labels[i]
        payload = allreduce(payload, "sum")
        old_id = int(round(payload[3]))
        sums[old_id] -= payload[:3]
        sums[new_id] = payload[:3]
        counts[new_id] = 1.0
        counts[old_id] -= 1.0


_cdf_cache = {}


def _uniform_cdf(n):
    """The normalised running sum ``RandomState.choice(n, p=ones/n)`` searches (kept for the last cloud size: frames
    of a stream usually repeat it, and the four passes over n doubles cost as much as 25 Lloyd iterations at 1e5)."""
    if n not in _cdf_cache:
        _cdf_cache.clear()
        w = np.ones(n)
        cdf = (w / w.sum()).cumsum()
        cdf /= cdf[-1]
        _cdf_cache[n] = cdf
    return _cdf_cache[n]


def column_mean_var(X):
    """``X.mean(axis=0)``, ``np.var(X, axis=0)`` and ``X - mean`` of an [N,3] float64 array, bit for bit, at memory
    speed: NumPy reduces a C-ordered [N,3] array over axis 0 row by row (a plain running sum per column, 3 elements
    per inner loop -- 25 ms per million points for the variance alone); the library's host helper
    ``hgmm_kmeans_center_f64`` runs the same three add chains in ~3 ms.  (tests/test_kmeans_cpu.py holds the
    equality.)"""
    import ctypes as C
    from ._native import load_library
    Xd = np.ascontiguousarray(X, dtype=np.float64)
    mean, var, Xc = np.empty(3), np.empty(3), np.empty_like(Xd)
    rc = load_library().hgmm_kmeans_center_f64(Xd.ctypes.data_as(C.c_void_p), len(Xd), mean.ctypes.data_as(C.c_void_p),
                                               var.ctypes.data_as(C.c_void_p), Xc.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError("hgmm_kmeans_center_f64 failed (%d) on an array of shape %s" % (rc, X.shape))
    return mean, var, Xc


class KMeans:
    def __init__(self, n_clusters=8, random_state=None, max_iter=300, n_init=1, tol=1e-4,
                 init="k-means++", ctx: Context | None = None):
        if n_init != 1:
            raise ValueError("only n_init=1 (what the reference uses) is supported")
        if isinstance(init, str) and init != "k-means++":
            raise ValueError("init must be 'k-means++' or an array of centres")
        self.init = init
        self.n_clusters = int(n_clusters)
        self.random_state = random_state
        self.max_iter = int(max_iter)
        self.n_init = 1
        self.tol = float(tol)
        self._ctx = ctx

    # -- seeding -------------------------------------------------------------------------
    def _seed(self, ctx, n, rs):
        k = self.n_clusters
        trials = 2 + int(np.log(k))
        # == rs.choice(n, p=w / w.sum()) with w = 1 (the draw scikit-learn makes), without choice()'s
        # validation passes over p: one uniform against the normalised running sum of p
        first = int(_uniform_cdf(n).searchsorted(rs.random_sample(), side='right'))
        rand = rs.uniform(size=(max(k - 1, 0), trials))       # == k-1 successive draws of `trials`
        import time
        t0 = time.perf_counter()
        ids, centres = ctx.kmeans_plusplus(k, first, rand)
        self.seeding_ms_ = (time.perf_counter() - t0) * 1e3     # device k-means++ incl. the transfer of the draws
        return ids, centres

    # -- scikit-learn's _relocate_empty_clusters_dense ---------------------------------------
    @staticmethod
    def _relocate_empty(ctx, Xc, sums, counts):
        empty = np.where(counts == 0)[0]
        if len(empty) == 0:
            return
        labels, dist = ctx.kmeans_labels(with_distances=True)
        if getattr(ctx, "nranks", 1) > 1:
            relocate_empty_sharded(lambda v, op: ctx.allreduce(v, op=op), ctx.rank, labels, dist, Xc, sums, counts)
            return
        far = np.argpartition(dist, -len(empty))[:-len(empty) - 1:-1]
        for new_id, far_idx in zip(empty, far):
            old_id = labels[far_idx]
            sums[old_id] -= Xc[far_idx]
            sums[new_id] = Xc[far_idx]
            counts[new_id] = 1.0
            counts[old_id] -= 1.0

    def fit(self, X):
        X = np.asarray(X.points if hasattr(X, "points") else X, dtype=np.float64)
        if X.ndim != 2 or X.shape[1] != 3:
            raise ValueError("expected an [N,3] array, got %s" % (X.shape,))
        n, k = len(X), self.n_clusters
        if n < k:
            raise ValueError("n_samples=%d should be >= n_clusters=%d." % (n, k))
        ctx = self._ctx or default_context()
        ranks = getattr(ctx, "nranks", 1)
        if ranks > 1:
            # X is this rank's shard of one joint clustering: global mean / variance / count
            tot = ctx.allreduce(np.concatenate([[float(n)], X.sum(axis=0), (X * X).sum(axis=0)]))
            mean = tot[1:4] / tot[0]
            var = tot[4:7] / tot[0] - mean * mean
            Xc = X - mean
        else:
            mean, var, Xc = column_mean_var(X)
        tol_abs = 0.0 if self.tol == 0 else float(np.mean(var) * self.tol)
        ctx.set_points(Xc)                            # (the context's own cloud; DevicePoints handles stay valid)
        if isinstance(self.init, str):
            rs = _random_state(self.random_state)
            self.init_indices_, centres = self._seed(ctx, n, rs)
            if ranks > 1:                             # seeds come from rank 0's shard (sum = broadcast)
                centres = ctx.allreduce(centres if ctx.rank == 0 else np.zeros_like(centres))
        else:
            centres = np.array(self.init, dtype=np.float64).reshape(k, 3) - mean
            self.init_indices_ = None

        strict = False
        n_iter = 0
        inertia = 0.0
        first = True
        while n_iter < self.max_iter:
            pending = None
            if ranks == 1:
                # the loop runs on the device until a stop rule fires or a cluster comes up empty
                centres, done_it, strict, pending = ctx.kmeans_lloyd(centres, self.max_iter - n_iter, tol_abs,
                                                                     reset_labels=first)
                n_iter += done_it
                first = False
                if pending is None:
                    break
                sums, counts, changed = pending
            else:
                sums, counts, inertia, changed = ctx.kmeans_step(centres, reset_labels=first)
                first = False
            # one iteration finished on the host (always under a communicator; after an empty cluster otherwise)
            self._relocate_empty(ctx, Xc, sums, counts)
            new = sums
            pos = counts > 0
            new[pos] *= (1.0 / counts[pos])[:, None]
            shift_tot = float((np.sqrt(((new - centres) ** 2).sum(axis=1)) ** 2).sum())
            centres = new
            n_iter += 1
            if changed == 0:
                strict = True
                break
            if shift_tot <= tol_abs:
                break
        if not strict:
            _, _, inertia, _ = ctx.kmeans_step(centres)
        self.labels_ = ctx.kmeans_labels()
        if strict:
            # labels are those of the last assignment; inertia is measured against the final centres
            inertia = float(((Xc - centres[self.labels_]) ** 2).sum())
            if ranks > 1:
                inertia = float(ctx.allreduce([inertia])[0])
        self.inertia_ = float(inertia)
        self.n_iter_ = n_iter
        self.cluster_centers_ = centres + mean
        return self

    def fit_predict(self, X):
        return self.fit(X).labels_


def kmeans_centres(X, k, random_state=1, max_iter=50, ctx: Context | None = None):
    """``KMeans(n_clusters=k, random_state=1, max_iter=50, n_init=1).fit(X).cluster_centers_``
    (gmmreg_gpu/gmm_impl.py:20-21) on the device."""
    return KMeans(n_clusters=k, random_state=random_state, max_iter=max_iter, n_init=1, ctx=ctx).fit(X).cluster_centers_
