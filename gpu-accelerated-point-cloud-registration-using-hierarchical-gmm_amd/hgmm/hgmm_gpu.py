"""Drop-in for the reference's ``src/python/hgmm/hgmm_gpu.py`` (and its CPU twin
``hgmm_cupy_cpu_working.py``): ``buildGMMTree``, ``GMMTree``, ``registration_gmmtree``,
``RigidTransformation`` with the same signatures and return values.

What runs where
  * tree construction (E-step, M-step, level log-likelihood, the per-level partition) and the
    registration E-step (tree descent + moment accumulation): HIP kernels, float64
    (csrc/tree_kernels.hip) -- the reference ran the latter as a pure-Python triple loop
    (hgmm_gpu.py:550-577);
  * the rigid update (per-node eigh + 6-dof twist least squares, hgmm_gpu.py:729-752) stays on
    the host in NumPy, as in the reference.

Semantics follow the CPU twin where the two reference files differ (empty-node rule m0 < ld;
the Numba file has it commented out, hgmm_gpu.py:250-256).  Initialisation constants default
to the Numba file's (seed 72, sig2 = 0.004, hgmm_gpu.py:469-477) and can be overridden.
"""
from __future__ import annotations

import contextlib
import time
from collections import namedtuple

import numpy as np

from .._native import Context, default_context

eps = np.float32(1.0e-15)     # hgmm_gpu.py:29
n_node = 8                    # hgmm_gpu.py:30


def child2(j):
    return (j + 1) * n_node


def level(l):
    """First node index of level l (hgmm_gpu.py:91-92)."""
    return n_node * (np.power(n_node, l) - 1) / (n_node - 1)


def n_total_nodes(maxTreeLevel):
    return int(n_node * (np.power(n_node, maxTreeLevel) - 1) / (n_node - 1))


def complexity(cov):
    """smallest eigenvalue / trace (hgmm_gpu.py:78-82)."""
    lam = np.linalg.eigvalsh(np.asarray(cov, dtype=np.float64))
    return lam[0] / np.sum(lam)


def _points(x):
    return np.asarray(x.points if hasattr(x, "points") else x)


def buildGMMTree(points, maxTreeLevel, ls, ld, sig2=0.004, seed=72, init_idx=None,
                 max_iters_per_level=1000, ctx: Context | None = None, return_trace=False, dtype=None, pdf_dtype=None):
    """-> (mixingCoeff[T], mean[T,3], covar[T,3,3])   (hgmm_gpu.py:466-548).

    ``init_idx`` (T indices into ``points``) overrides the reference's
    ``np.random.seed(72); randint(nTotal, size=nTotal)`` draw.

    ``dtype``: the reference has this function twice -- float64 in the CPU twin (hgmm_cupy_cpu_working.py:122-160, the
    canonical semantics) and float32 in the GPU file (``points.astype(np.float32)``, float32 node and moment arrays,
    hgmm_gpu.py:472, 478-484).  Default: the points' own type -- float32 points give float32 tables and the float32-pdf
    stop rule (``Context.tree_set_precision``), anything else the float64 path.  ``pdf_dtype`` (default: ``dtype``) sets the
    stop rule's arithmetic alone: :class:`GMMTree` keeps float64 tables for its registration whatever the points' type
    (``dtype=np.float64, pdf_dtype=<the points' type>``).  The context's own precision setting is restored afterwards."""
    ctx = ctx or default_context()
    raw = _points(points)
    dt = np.dtype(dtype) if dtype is not None else (np.dtype(np.float32) if raw.dtype == np.float32 else np.dtype(np.float64))
    pdt = np.dtype(pdf_dtype) if pdf_dtype is not None else dt
    if dt not in (np.dtype(np.float32), np.dtype(np.float64)) or pdt not in (np.dtype(np.float32), np.dtype(np.float64)):
        raise ValueError("dtype must be float32 or float64")
    P = np.ascontiguousarray(raw, dtype=np.float64)
    T = n_total_nodes(maxTreeLevel)
    if init_idx is None:
        rs = np.random.RandomState(seed)
        init_idx = rs.randint(T, size=T)
    ctx.set_points(P)
    prev = getattr(ctx, "tree_dtype", np.dtype(np.float64))     # (a precision the caller set on the context survives this call)
    ctx.tree_set_precision(pdt)
    try:
        pi, mu, cov, leaf, iters, q = ctx.tree_build(maxTreeLevel, ls, ld, P[np.asarray(init_idx)], sig2,
                                                     max_iters_per_level, want_leaf=bool(return_trace))
    finally:
        ctx.tree_set_precision(prev)
    if dt == np.dtype(np.float32):
        pi, mu, cov = pi.astype(np.float32), mu.astype(np.float32), cov.astype(np.float32)
    if return_trace:
        return pi, mu, cov, {"leaf_idx": leaf, "iters_per_level": iters, "q_trace": q}
    return pi, mu, cov


def gmmTreeEStep(points, mixingCoeff, mean, covar, parentIdx, ctx: Context | None = None):
    """One tree E-step for an arbitrary parent assignment (hgmm_cupy_cpu_working.py:162-191).
    -> (momentsZero[T], momentsOne[T,3], momentsTwo[T,3,3], currentIdx[N])."""
    ctx = ctx or default_context()
    ctx.set_points(np.ascontiguousarray(_points(points), dtype=np.float64))
    return ctx.tree_estep(mixingCoeff, mean, covar, parentIdx)


def gmmTreeMStep(momentsZero, momentsOne, momentsTwo, l, mixingCoeff, mean, covar, n_points, ld,
                 ctx: Context | None = None):
    """ML update of the nodes of level ``l`` (hgmm_cupy_cpu_working.py:193-198 with the m0 < ld
    rule of mlEstimator 109-119).  -> new (mixingCoeff, mean, covar)."""
    ctx = ctx or default_context()
    return ctx.tree_mstep(momentsZero, momentsOne, momentsTwo, int(level(l)), int(level(l + 1)), n_points, ld,
                          mixingCoeff, mean, covar)


def logLikelihoodValue(mixingCoeff, mean, covar, data, j0, j1, ctx: Context | None = None):
    """sum_i log max(sum_{j0 <= j < j1, pi_j >= eps} pi_j N(x_i; j), eps)
    (hgmm_cupy_cpu_working.py:72-85 / the Numba kernel hgmm_gpu.py:107-115 + sum_reduce)."""
    ctx = ctx or default_context()
    ctx.set_points(np.ascontiguousarray(_points(data), dtype=np.float64))
    return ctx.tree_loglik(mixingCoeff, mean, covar, int(j0), int(j1))


def fitFullCovGMM(points, n_components, ls=80.0, ld=1.0e-4, sig2=0.00034, init_idx=None, seed=None,
                  max_iters=1000, ctx: Context | None = None, return_trace=False, dtype=None):
    """Flat full-covariance GMM = ONE tree level with branching ``n_components`` (the reference's
    CPU twin run with its module global ``n_node = J`` and ``buildGMMTree(P, 1, ls, ld)``,
    hgmm_cupy_cpu_working.py:30,122-160).  -> (mixingCoeff[J], mean[J,3], covar[J,3,3]).

    ``init_idx``: J indices of the initial means (default: the twin's draw
    ``RandomState(J).randint(J, size=J)`` -- note it only ever picks among the first J points).

    ``dtype=np.float32`` (explicit only; the points' own type does NOT switch it): the float32-tile kernel for this fit
    (``Context.tree_set_precision``; J <= 1024) -- float32 pdfs, tile and matrix-core statistics about the centroid, float64
    row sums / log / M-step; ~25 % faster per iteration at 10^6 x 800, Sigma to ~1e-5 of a variance.  The context's own
    setting is restored afterwards."""
    ctx = ctx or default_context()
    P = np.ascontiguousarray(_points(points), dtype=np.float64)
    J = int(n_components)
    if init_idx is None:
        init_idx = np.random.RandomState(J if seed is None else seed).randint(J, size=J)
    ctx.set_points(P)
    prev = getattr(ctx, "tree_dtype", np.dtype(np.float64))
    if dtype is not None:
        ctx.tree_set_precision(dtype)
    try:
        pi, mu, cov, labels, q = ctx.fullcov_fit(J, ls, ld, P[np.asarray(init_idx)], sig2, max_iters)
    finally:
        if dtype is not None:
            ctx.tree_set_precision(prev)
    if return_trace:
        return pi, mu, cov, {"labels": labels, "q_trace": q}
    return pi, mu, cov


def gmmTreeRegESTep(points, mixingCoeff, mean, covar, maxTreeLevel, lc, ctx: Context | None = None):
    """-> (momentsZero[T], momentsOne[T,3], momentsTwo[T,3,3])   (hgmm_gpu.py:550-577)."""
    ctx = ctx or default_context()
    ctx.tree_set_nodes(maxTreeLevel, mixingCoeff, mean, covar)
    ctx.tree_set_target(_points(points))
    return ctx.tree_reg_estep(n_total_nodes(maxTreeLevel), lambda_c=lc)


# RigidTransformation (hgmm_gpu.py:599-618: x -> scale R x + t on row vectors, ``inverse()``) is the same
# object the GMMReg mirror uses
from ..gmmreg_gpu.transforms import Transformation, RigidTransformation  # noqa: E402,F401


def skew(x):
    """[x]_x, the cross-product matrix (hgmm_gpu.py:620-624)."""
    a, b, c = x
    return np.array([[0.0, -c, b], [c, 0.0, -a], [-b, a, 0.0]])


def twist_trans(tw, linear=False):
    """6-twist (omega, v) -> (R, t): exp of omega by Rodrigues' formula, or its first-order form I + [omega]_x with
    ``linear`` (hgmm_gpu.py:646-664).  The translation part is passed through as is."""
    omega, v = np.asarray(tw[:3], dtype=np.float64), tw[3:]
    if linear:
        return np.identity(3) + skew(omega), v
    angle = np.linalg.norm(omega)
    if angle == 0.0:
        return np.identity(3), v
    axis = omega / angle
    k = skew(axis)
    return np.identity(3) + np.sin(angle) * k + (1.0 - np.cos(angle)) * (k @ k), v


def twist_mul(tw, rot, t, linear=False):
    """Compose the twist's motion with (rot, t): (dR rot, dR t + dt)   (hgmm_gpu.py:634-644)."""
    d_rot, d_t = twist_trans(tw, linear=linear)
    return d_rot @ rot, d_rot @ t + d_t


_tp_controller = None


def _small_lapack():
    """The M-step's factorizations are tall-and-skinny (3 n_nodes x 7): a threaded LAPACK spends
    its time synchronising (16 ms on 8 threads vs 0.6 ms on one for 9900 x 7), so they run on one
    thread when threadpoolctl is there to say so.  The controller is built once (inspecting the
    loaded libraries on every call costs more than the factorization)."""
    global _tp_controller
    try:
        if _tp_controller is None:
            from threadpoolctl import ThreadpoolController
            _tp_controller = ThreadpoolController()
        return _tp_controller.limit(limits=1, user_api="blas")
    except Exception:                                   # pragma: no cover - optional dependency
        import contextlib
        return contextlib.nullcontext()


EstepResult = namedtuple('EstepResult', ['momentZero', 'momentOne', 'momentTwo'])
MstepResult = namedtuple('MstepResult', ['transformation', 'q'])


class GMMTree():
    """GMM tree registration (hgmm_gpu.py:669-768).

    Args mirror the reference; ``ls`` / ``ld`` / ``sig2`` / ``init_idx`` expose the constants it
    hard-codes (20, 1e-4, 0.004, seed 72)."""

    def __init__(self, source=None, tree_level=5, lambda_c=0.01, ls=20, ld=1.0e-4, sig2=0.004,
                 init_idx=None, ctx: Context | None = None, verbose=False, solve_on_device=False):
        self._source = None
        self._tree_level = tree_level
        self._lambda_c = lambda_c
        self._ls, self._ld, self._sig2, self._init_idx = ls, ld, sig2, init_idx
        self._tf_type = RigidTransformation
        self._tf_result = self._tf_type()
        self._callbacks = []
        self._ctx_arg = ctx
        self._verbose = verbose
        self._device_mstep = True            # False: every iteration through expectation_step + maximization_step
        # True: the library's loop also solves the 6 x 6 system, composes the twist and applies the stop rule on the device
        # (per-context option reg_device_solve; the host solve is the default and the parity reference)
        self._solve_on_device = bool(solve_on_device)
        self._target_id = None
        if source is not None:
            self.set_source(source)

    @property
    def _ctx(self):
        if self._ctx_arg is None:
            self._ctx_arg = default_context()
        return self._ctx_arg

    def set_source(self, source):
        self._source = _points(source)
        self._eig_cache = None                       # node covariances are about to change
        t1 = time.time()
        # float64 tables whatever the points' type: the registration re-uploads them, takes eigh of the covariances and
        # 1 / sqrt of their eigenvalues -- a thin leaf covariance rounded to float32 can turn indefinite.  The
        # stop rule's arithmetic still follows the points' type, as in buildGMMTree.
        pdf = np.float32 if self._source.dtype == np.float32 else np.float64
        self._mixingCoeff, self._mean, self._covar = buildGMMTree(
            self._source, self._tree_level, self._ls, self._ld, sig2=self._sig2,
            init_idx=self._init_idx, ctx=self._ctx, dtype=np.float64, pdf_dtype=pdf)
        if self._verbose:
            print("Build tree Time: ", time.time() - t1)

    def set_nodes(self, mixingCoeff, mean, covar):
        """Use an existing tree (e.g. a saved one) instead of building from a source cloud."""
        self._mixingCoeff = np.asarray(mixingCoeff, dtype=np.float64)
        self._mean = np.asarray(mean, dtype=np.float64)
        self._covar = np.array(covar, dtype=np.float64)        # own copy: the eigh cache below belongs to it
        self._eig_cache = None

    def set_callbacks(self, callbacks):
        self._callbacks = callbacks

    def expectation_step(self, target=None):
        """With ``target`` given: E-step on that (already transformed) cloud, like the reference.
        Inside :meth:`registration` the resident target is transformed on the device instead."""
        T = len(self._mixingCoeff)
        if target is not None:
            self._ctx.tree_set_target(_points(target))
            self._target_id = None
            m = self._ctx.tree_reg_estep(T, lambda_c=self._lambda_c)
        else:
            tf = self._tf_result
            m = self._ctx.tree_reg_estep(T, tf.rot, tf.t, tf.scale, self._lambda_c)
        return EstepResult(*m)

    def _node_eig(self):
        """eigh of every node covariance.  The nodes do not change during a registration, so the
        decomposition the reference recomputes per node and per iteration (hgmm_gpu.py:737) is
        done once per tree."""
        if getattr(self, "_eig_cache", None) is None:     # invalidated by set_source / set_nodes
            self._eig_cache = np.linalg.eigh(self._covar)
        return self._eig_cache

    def maximization_step(self, estep_res, trans_p):
        """Twist least squares over the nodes that received mass (hgmm_gpu.py:729-752).

        Same system as the reference's ``np.linalg.lstsq(Amat, bmat)`` with its all-zero rows (nodes
        without mass) left out, solved through one QR of [A | b]: x = R^-1 (Q^T b), q = squared
        residual = R[6,6]^2.  Rank-deficient or tiny systems take the reference's exact call."""
        m0, m1 = estep_res.momentZero, estep_res.momentOne
        n = len(self._mixingCoeff)
        live = np.nonzero(~(m0 < np.finfo(np.float32).eps))[0]
        x = q = None
        if len(live) > 2:
            lmd_all, nn_all = self._node_eig()
            lmd, nn = lmd_all[live], nn_all[live]
            s = m1[live] / m0[live][:, None]
            nn = nn * np.sqrt(m0[live][:, None] / lmd)[:, None, :]
            nnT = np.transpose(nn, (0, 2, 1))
            ab = np.empty((3 * len(live), 7))
            ab[:, :3] = np.cross(s[:, None, :], nnT).reshape(-1, 3)
            ab[:, 3:6] = nnT.reshape(-1, 3)
            ab[:, 6] = (np.einsum('nij,nj->ni', nnT, self._mean[live]) - np.einsum('nij,nj->ni', nnT, s)).ravel()
            if np.isfinite(ab).all():
                with _small_lapack():
                    r = np.linalg.qr(ab, mode='r')
                d = np.abs(np.diag(r)[:6])
                if d.min() > 1e-12 * d.max():
                    x = np.linalg.solve(r[:6, :6], r[:6, 6])
                    q = np.array([r[6, 6] ** 2])
        if x is None:
            amat = np.zeros((n * 3, 6))
            bmat = np.zeros(n * 3)
            if len(live):
                lmd, nn = np.linalg.eigh(self._covar[live])
                s = m1[live] / m0[live][:, None]
                nn = nn * np.sqrt(m0[live][:, None] / lmd)[:, None, :]
                nnT = np.transpose(nn, (0, 2, 1))
                b = np.einsum('nij,nj->ni', nnT, self._mean[live]) - np.einsum('nij,nj->ni', nnT, s)
                rows = (3 * live[:, None] + np.arange(3)[None, :]).ravel()
                bmat[rows] = b.ravel()
                amat[rows, :3] = np.cross(s[:, None, :], nnT).reshape(-1, 3)
                amat[rows, 3:] = nnT.reshape(-1, 3)
            x, q, _, _ = np.linalg.lstsq(amat, bmat, rcond=-1)
        rot, t = twist_mul(x, trans_p.rot, trans_p.t)
        return MstepResult(RigidTransformation(rot, t), q)

    def _device_iteration(self):
        """One registration iteration with E-step AND the M-step's least-squares system on the device: the host
        receives the 6 x 6 normal equations (28 numbers), solves them and composes the twist.  The system is the
        reference's (hgmm_gpu.py:729-752) -- sum_c n_c n_c^T = m0 Sigma^-1 makes the per-node eigh unnecessary.
        Returns None when the system is too ill-conditioned for normal equations (the caller then takes the
        reference's stacked least-squares path)."""
        tf = self._tf_result
        ata, atb, btb = self._ctx.tree_reg_normal(tf.rot, tf.t, tf.scale, self._lambda_c)
        if not (np.isfinite(ata).all() and np.isfinite(atb).all()):
            return None
        lam = np.linalg.eigvalsh(ata)
        if not lam[-1] > 0.0 or lam[0] <= 1e-11 * lam[-1]:
            return None
        x = np.linalg.solve(ata, atb)
        q = np.array([max(btb - float(x @ atb), 0.0)])
        rot, t = twist_mul(x, tf.rot, tf.t)
        return MstepResult(RigidTransformation(rot, t), q)

    def _registration_in_library(self, maxiter, tol, _resume=None):
        """The loop of :meth:`registration` without per-iteration Python: ``hgmm_tree_register`` iterates (E-step and
        normal equations on the device, 6 x 6 solve / twist / stop rule on the library's host side) until it stops or
        meets a system too ill-conditioned for normal equations; that one iteration is then done here with the
        reference's stacked least squares, and the library carries on.  Same arithmetic per iteration as
        :meth:`_device_iteration` (used when callbacks want every intermediate transformation)."""
        tf = self._tf_result
        rot, t = np.asarray(tf.rot, dtype=np.float64), np.asarray(tf.t, dtype=np.float64)
        q = None
        it = 0
        host_step_due = False
        if _resume is not None:                      # (a pair that left a batch at status 2: ``it`` iterations done, last q)
            it, q, host_step_due = _resume
        while it < maxiter:
            if not host_step_due:
                with (self._ctx.config(reg_device_solve=1) if self._solve_on_device else contextlib.nullcontext()):
                    rot, t, done, q_new, status, _ = self._ctx.tree_register(rot, t, tf.scale, self._lambda_c, maxiter - it, tol, q)
                it += done
                if done:
                    q = q_new
                self._tf_result = RigidTransformation(rot, t, tf.scale)
                if status != 2:
                    break
            host_step_due = False
            res = self.maximization_step(self.expectation_step(), self._tf_result)      # host M-step, one iteration
            self._tf_result = res.transformation
            rot, t = np.asarray(res.transformation.rot, dtype=np.float64), np.asarray(res.transformation.t, dtype=np.float64)
            it += 1
            q_host = float(np.ravel(res.q)[0]) if np.size(res.q) else None
            if q is not None and q_host is not None and abs(q_host - q) < tol:
                q = q_host
                break
            q = q_host
        self.n_iter_ = it
        return MstepResult(self._tf_result.inverse(), np.array([q]) if q is not None else np.array([]))

    def registration(self, target, maxiter=20, tol=1.0e-4):
        """-> MstepResult(tf.inverse(), q)   (hgmm_gpu.py:754-768)."""
        self._ctx.tree_set_nodes(self._tree_level, self._mixingCoeff, self._mean, self._covar)
        self._ctx.tree_set_target(_points(target))
        if self._device_mstep and not self._callbacks:
            return self._registration_in_library(maxiter, tol)
        q = None
        res = None
        self.n_iter_ = 0
        for _ in range(maxiter):
            self.n_iter_ += 1
            res = self._device_iteration() if self._device_mstep else None
            if res is None:
                estep_res = self.expectation_step()      # target transformed on the device
                res = self.maximization_step(estep_res, self._tf_result)
            self._tf_result = res.transformation
            for c in self._callbacks:
                c(self._tf_result.inverse())
            if q is not None and np.size(q) and np.size(res.q) and abs(res.q - q) < tol:
                break
            q = res.q
        return MstepResult(self._tf_result.inverse(), res.q)


def euler_matrix_xyz(ai, aj, ak):
    """Static-frame x-y-z Euler angles -> 3x3 rotation (what the absent third-party
    ``transformations.euler_matrix(ai, aj, ak)`` default 'sxyz' returns; hgmm_gpu.py:790)."""
    ci, si, cj, sj, ck, sk = np.cos(ai), np.sin(ai), np.cos(aj), np.sin(aj), np.cos(ak), np.sin(ak)
    rx = np.array([[1, 0, 0], [0, ci, -si], [0, si, ci]])
    ry = np.array([[cj, 0, sj], [0, 1, 0], [-sj, 0, cj]])
    rz = np.array([[ck, -sk, 0], [sk, ck, 0], [0, 0, 1.0]])
    return rz @ ry @ rx


def prepare_source_and_target_rigid_3d(source, noise_amp=0.001, n_random=500,
                                       orientation=np.deg2rad([0.0, 0.0, 30.0]), translation=np.zeros(3),
                                       voxel_size=0.005, rng=None):
    """Noisy, cluttered, rigidly moved copy of a cloud for registration experiments
    (hgmm_gpu.py:772-795 minus Open3D/normals): voxel down-sample, shuffle, add Gaussian noise and
    ``n_random`` uniform outliers in a 1.5x bounding box, then rotate/translate.
    ``source``: [N,3] array or a .ply/.pcd path.  -> (source_points, target_points)."""
    from ..pointcloud_io import read_point_cloud, voxel_down_sample
    rng = rng or np.random
    src = read_point_cloud(source) if isinstance(source, str) else np.asarray(source, dtype=np.float64)
    if voxel_size:
        src = voxel_down_sample(src, voxel_size)
    tp = src.copy()
    rng.shuffle(tp)
    rg = 1.5 * (tp.max(axis=0) - tp.min(axis=0))
    rands = (rng.rand(n_random, 3) - 0.5) * rg + tp.mean(axis=0)
    tgt = np.r_[tp + noise_amp * rng.randn(*tp.shape), rands]
    rot = euler_matrix_xyz(*orientation)
    return src, tgt @ rot.T + np.asarray(translation)


def registration_gmmtree(source, target, maxiter=20, tol=1.0e-4, callbacks=[], **kargs):
    """hgmm_gpu.py:802-807."""
    gt = GMMTree(_points(source), **kargs)
    gt.set_callbacks(callbacks)
    return gt.registration(_points(target), maxiter, tol)


BATCH_MAX_POINTS = 400000       # hgmm_tree_build_batch takes clouds below this size (csrc/tree_batch.hip)


def registration_gmmtree_batch(pairs, maxiter=20, tol=1.0e-4, ctx: Context | None = None, tree_level=5, lambda_c=0.01,
                               ls=20, ld=1.0e-4, sig2=0.004, init_idx=None, return_info=False, pdf_dtype=None,
                               solve_on_device=False):
    """``[registration_gmmtree(s, t, maxiter, tol, tree_level=..., ...) for s, t in pairs]`` (hgmm_gpu.py:802-807 per pair)
    with ALL pairs in the same launches: the B source clouds are one resident forest (``hgmm_tree_build_batch``: levels in
    lock-step, one stop rule per cloud), the B targets are registered against their trees together
    (``hgmm_tree_register_batch``: one E-step + one normal-equations launch per iteration, the 6 x 6 solves on the host).
    Every pair's tree, iteration counts and transformation are bitwise what the serial call returns for it.  A pair whose
    normal equations turn ill-conditioned leaves the batch and is finished by the serial path (stacked least squares on
    the host, like :meth:`GMMTree._registration_in_library`).  Clouds of 400 000 points or more go through the serial call.

    ``pdf_dtype``: the arithmetic of the build's stop rule, as in :class:`GMMTree` -- default: each SOURCE's own type
    (float32 scans, the type of the reference's GPU file, hgmm_gpu.py:472: float32 pdfs; anything else float64); pairs
    of either kind in one call run as two batches.  Tables, E-step, moments and the registration are float64 either way.

    -> list of ``MstepResult(transformation, q)`` in the order of ``pairs`` (+ a dict with the per-pair build / registration
    iteration counts with ``return_info``)."""
    ctx = ctx or default_context()
    pairs = list(pairs)
    if not pairs:
        return ([], {}) if return_info else []
    big = [k for k, (s, _) in enumerate(pairs) if len(_points(s)) >= BATCH_MAX_POINTS]
    if big:
        # a cloud of >= 400 000 points fills the chip by itself and takes the serial build's four-points-per-thread
        # log-likelihood, which the batched path does not reproduce: those pairs run one by one, the rest as a batch
        out = [None] * len(pairs)
        info = {"build_iters": np.full((len(pairs), tree_level), -1, np.int32), "registration_iters": [None] * len(pairs),
                "status": [None] * len(pairs)}
        for k in big:
            gt = GMMTree(pairs[k][0], tree_level=tree_level, lambda_c=lambda_c, ls=ls, ld=ld, sig2=sig2, init_idx=init_idx, ctx=ctx,
                         solve_on_device=solve_on_device)
            out[k] = gt.registration(_points(pairs[k][1]), maxiter, tol)
            info["registration_iters"][k], info["status"][k] = int(gt.n_iter_), 0
        rest = [k for k in range(len(pairs)) if k not in set(big)]
        if rest:
            r, inf = registration_gmmtree_batch([pairs[k] for k in rest], maxiter, tol, ctx, tree_level, lambda_c, ls, ld, sig2,
                                                init_idx, True, pdf_dtype, solve_on_device)
            for j, k in enumerate(rest):
                out[k] = r[j]
                info["build_iters"][k] = inf["build_iters"][j]
                info["registration_iters"][k], info["status"][k] = inf["registration_iters"][j], inf["status"][j]
        return (out, info) if return_info else out
    if pdf_dtype is None:
        kinds = [np.dtype(np.float32) if _points(s).dtype == np.float32 else np.dtype(np.float64) for s, _ in pairs]
        if len(set(kinds)) > 1:                                   # one batch per kind, results back in the caller's order
            out, info = [None] * len(pairs), {"build_iters": [None] * len(pairs), "registration_iters": [None] * len(pairs),
                                              "status": [None] * len(pairs)}
            for kind in sorted(set(kinds), key=str):
                sel = [k for k, v in enumerate(kinds) if v == kind]
                r, inf = registration_gmmtree_batch([pairs[k] for k in sel], maxiter, tol, ctx, tree_level, lambda_c, ls, ld,
                                                    sig2, init_idx, True, kind, solve_on_device)
                for j, k in enumerate(sel):
                    out[k] = r[j]
                    for key in info:
                        info[key][k] = inf[key][j]
            info["build_iters"] = np.asarray(info["build_iters"])
            return (out, info) if return_info else out
        pdf_dtype = kinds[0]
    clock = [time.perf_counter()]
    # (float32 clouds are handed over as they are -- the library widens them on the device, exactly -- and only the T rows of
    #  the initial means are widened here)
    srcs = [_points(s) for s, _ in pairs]
    tgts = [_points(t) for _, t in pairs]
    B = len(pairs)
    T = n_total_nodes(tree_level)
    idx = np.asarray(init_idx) if init_idx is not None else np.random.RandomState(72).randint(T, size=T)
    init_mu = np.stack([np.asarray(S[idx], dtype=np.float64) for S in srcs])
    clock.append(time.perf_counter())
    ctx.set_points_batch(srcs)
    clock.append(time.perf_counter())
    prev = getattr(ctx, "tree_dtype", np.dtype(np.float64))     # (a precision the caller set on the context survives this call)
    ctx.tree_set_precision(pdf_dtype)
    try:
        _, build_iters, _ = ctx.tree_build_batch([len(S) for S in srcs], tree_level, ls, ld, init_mu, sig2, want_tables=False)
    finally:
        ctx.tree_set_precision(prev)
    clock.append(time.perf_counter())
    ctx.tree_set_targets_batch(tgts)
    clock.append(time.perf_counter())
    rot0 = np.tile(np.identity(3), (B, 1, 1))
    if solve_on_device:                                          # (per-context option reg_device_solve for this call)
        with ctx.config(reg_device_solve=1):
            rot, t, iters, q, status, _ = ctx.tree_register_batch(rot0, np.zeros((B, 3)), 1.0, lambda_c, maxiter, tol)
    else:
        rot, t, iters, q, status, _ = ctx.tree_register_batch(rot0, np.zeros((B, 3)), 1.0, lambda_c, maxiter, tol)
    clock.append(time.perf_counter())
    out = []
    reg_iters = [int(v) for v in iters]
    for b in range(B):
        if status[b] == 2:                                        # finish this pair through the serial entries
            gt = GMMTree(tree_level=tree_level, lambda_c=lambda_c, ls=ls, ld=ld, sig2=sig2, ctx=ctx, solve_on_device=solve_on_device)
            gt.set_nodes(*ctx.tree_get_nodes_batch(b, tree_level))
            gt._tf_result = RigidTransformation(rot[b], t[b])
            ctx.tree_set_nodes(tree_level, gt._mixingCoeff, gt._mean, gt._covar)
            ctx.tree_set_target(tgts[b])
            res = gt._registration_in_library(maxiter, tol, _resume=(int(iters[b]), None if np.isnan(q[b]) else float(q[b]), True))
            reg_iters[b] = int(gt.n_iter_)
            out.append(res)
            continue
        tf = RigidTransformation(rot[b].copy(), t[b].copy())
        out.append(MstepResult(tf.inverse(), np.array([q[b]]) if not np.isnan(q[b]) else np.array([])))
    if return_info:
        clock.append(time.perf_counter())
        # wall time of the call's phases, ms: host preparation (type conversion, initial means), upload of the sources, forest
        # build, upload of the targets, batched registration, result objects
        phases = dict(zip(("prepare", "sources_up", "build", "targets_up", "register", "results"),
                          (1e3 * np.diff(clock)).tolist()))
        return out, {"build_iters": build_iters, "registration_iters": reg_iters, "status": [int(v) for v in status],
                     "phases_ms": phases}
    return out
