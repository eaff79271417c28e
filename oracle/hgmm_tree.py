"""CPU oracle for the hierarchical-GMM (8-ary GMM tree) hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package may import this
module: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / the timed CPU baseline.

Vectorised NumPy (float64) restatement of the reference's *CPU twin*
``src/python/hgmm/hgmm_cupy_cpu_working.py`` (the canonical HGMM semantics; the
Numba file ``src/python/hgmm/hgmm_gpu.py`` has the empty-node guard commented
out and is not followed where the two differ).  No code is copied: the reference
is an object-per-node pure-Python triple loop, this is array code; every
function cites the lines it follows.

Pinning: ``tests/golden/hgmm_*.npz`` were produced by exec'ing the reference
itself (``tools/gen_golden.py``) and ``tests/test_oracle_golden.py`` checks this
module against them.
"""
from __future__ import annotations

from collections import namedtuple

import numpy as np

EPS = 1.0e-15           # hgmm_cupy_cpu_working.py:29
N_NODE = 8              # hgmm_cupy_cpu_working.py:30
TWO_PI_POW = (2.0 * np.pi) ** 1.5
F32_EPS = float(np.finfo(np.float32).eps)


def child(j, n_node=N_NODE):
    """First child of node j; root pseudo-parent is -1 (hgmm_cupy_cpu_working.py:93-94)."""
    return (j + 1) * n_node


def level(l, n_node=N_NODE):
    """Index of the first node of level l (hgmm_cupy_cpu_working.py:96-97)."""
    return n_node * (n_node ** l - 1) // (n_node - 1)


def n_total(max_level, n_node=N_NODE):
    return level(max_level, n_node)


# ---------------------------------------------------------------------------
# node tables: pi[T], mu[T,3], cov[T,3,3]   (object-per-node in the reference)
# ---------------------------------------------------------------------------

def det3(c):
    return (c[..., 0, 0] * (c[..., 1, 1] * c[..., 2, 2] - c[..., 1, 2] * c[..., 2, 1])
            - c[..., 0, 1] * (c[..., 1, 0] * c[..., 2, 2] - c[..., 1, 2] * c[..., 2, 0])
            + c[..., 0, 2] * (c[..., 1, 0] * c[..., 2, 1] - c[..., 1, 1] * c[..., 2, 0]))


def inv3(c, det):
    """Adjugate inverse of a batch of 3x3 matrices (det supplied)."""
    out = np.empty_like(c)
    out[..., 0, 0] = c[..., 1, 1] * c[..., 2, 2] - c[..., 1, 2] * c[..., 2, 1]
    out[..., 0, 1] = c[..., 0, 2] * c[..., 2, 1] - c[..., 0, 1] * c[..., 2, 2]
    out[..., 0, 2] = c[..., 0, 1] * c[..., 1, 2] - c[..., 0, 2] * c[..., 1, 1]
    out[..., 1, 0] = c[..., 1, 2] * c[..., 2, 0] - c[..., 1, 0] * c[..., 2, 2]
    out[..., 1, 1] = c[..., 0, 0] * c[..., 2, 2] - c[..., 0, 2] * c[..., 2, 0]
    out[..., 1, 2] = c[..., 0, 2] * c[..., 1, 0] - c[..., 0, 0] * c[..., 1, 2]
    out[..., 2, 0] = c[..., 1, 0] * c[..., 2, 1] - c[..., 1, 1] * c[..., 2, 0]
    out[..., 2, 1] = c[..., 0, 1] * c[..., 2, 0] - c[..., 0, 0] * c[..., 2, 1]
    out[..., 2, 2] = c[..., 0, 0] * c[..., 1, 1] - c[..., 0, 1] * c[..., 1, 0]
    return out / det[..., None, None]


def node_prep(cov):
    """Per-node quantities the reference recomputes per (point,node) pair
    (hgmm_cupy_cpu_working.py:62-70): validity (det >= eps), inverse, 1/(sqrt(det)(2pi)^1.5)."""
    det = det3(cov)
    ok = ~(det < EPS)
    safe = np.where(ok, det, 1.0)
    inv = inv3(cov, safe)
    coef = np.where(ok, 1.0 / (np.sqrt(safe) * TWO_PI_POW), 0.0)
    return ok, inv, coef


def pdf_pairs(x, mu, inv, coef):
    """N(x; mu, cov) for broadcastable batches; 0 where coef == 0 (det < eps).
    hgmm_cupy_cpu_working.py:62-70."""
    d = x - mu
    t = np.einsum('...i,...ij,...j->...', d, inv, d)
    return coef * np.exp(-0.5 * t)


def e_step(points, pi, mu, cov, parent_idx, lvl_nodes=None):
    """One tree E-step.  hgmm_cupy_cpu_working.py:162-191 (+accumulate 99-106).

    Returns (m0[T], m1[T,3], m2[T,3,3], current_idx[N], gamma[N,8]).
    """
    T = len(pi)
    n = len(points)
    j0 = child(np.asarray(parent_idx, dtype=np.int64))
    kid = j0[:, None] + np.arange(N_NODE)[None, :]                  # [N,8]
    ok, inv, coef = node_prep(cov)
    g = pi[kid] * pdf_pairs(points[:, None, :], mu[kid], inv[kid], coef[kid])
    den = g.sum(axis=1)
    good = den > EPS
    gamma = np.where(good[:, None], g / np.where(good, den, 1.0)[:, None], 0.0)
    cur = j0 + np.argmax(gamma, axis=1)
    use = np.where(gamma < EPS, 0.0, gamma)                          # accumulate() skips gamma < eps
    m0 = np.zeros(T)
    m1 = np.zeros((T, 3))
    m2 = np.zeros((T, 3, 3))
    flat = kid.ravel()
    np.add.at(m0, flat, use.ravel())
    np.add.at(m1, flat, (use[:, :, None] * points[:, None, :]).reshape(-1, 3))
    xx = points[:, :, None] * points[:, None, :]
    np.add.at(m2, flat, (use[:, :, None, None] * xx[:, None, :, :]).reshape(-1, 3, 3))
    return m0, m1, m2, cur, gamma


def m_step(m0, m1, m2, lvl, pi, mu, cov, n_points, ld):
    """In-place ML update of the nodes of level ``lvl``.
    hgmm_cupy_cpu_working.py:193-198 with mlEstimator 109-119 (m0 < ld -> pi=0, mu=0, cov=I)."""
    lb, le = level(lvl), level(lvl + 1)
    for j in range(lb, le):
        if m0[j] < ld:
            pi[j] = 0.0
            mu[j] = 0.0
            cov[j] = np.identity(3)
        else:
            pi[j] = m0[j] / n_points
            mu[j] = m1[j] / m0[j]
            cov[j] = m2[j] / m0[j] - np.outer(mu[j], mu[j])


def log_likelihood(points, pi, mu, cov, lvl, chunk=4096):
    """q = sum_i log max(sum_{j in level, pi_j >= eps} pi_j N(x_i; j), eps).
    hgmm_cupy_cpu_working.py:72-85."""
    lb, le = level(lvl), level(lvl + 1)
    sel = np.arange(lb, le)
    sel = sel[~(pi[sel] < EPS)]
    ok, inv, coef = node_prep(cov[sel])
    q = 0.0
    for s in range(0, len(points), chunk):
        x = points[s:s + chunk]
        if len(sel):
            p = pdf_pairs(x[:, None, :], mu[sel][None], inv[None], coef[None])
            tot = (p * pi[sel][None, :]).sum(axis=1)
        else:
            tot = np.zeros(len(x))
        q += np.log(np.maximum(tot, EPS)).sum()
    return q


def init_nodes(points, max_level, init_idx, sig2):
    """pi = 1/8, mu = points[idx], cov = sig2*I  (hgmm_cupy_cpu_working.py:123-136)."""
    T = n_total(max_level)
    pi = np.full(T, 1.0 / N_NODE)
    mu = np.array(points[np.asarray(init_idx)], dtype=np.float64)
    cov = np.tile(np.identity(3) * sig2, (T, 1, 1))
    return pi, mu, cov


BuildTrace = namedtuple('BuildTrace', ['q', 'iters_per_level', 'current_idx_per_level'])


def build_tree(points, max_level, ls, ld, init_idx, sig2=0.00034, max_iters_per_level=10000):
    """hgmm_cupy_cpu_working.py:122-160.  RNG stays outside (``init_idx`` explicit; the
    reference draws ``randint(nTotal, size=nTotal)`` from CuPy's generator).

    Returns (pi, mu, cov, trace)."""
    points = np.asarray(points, dtype=np.float64)
    pi, mu, cov = init_nodes(points, max_level, init_idx, sig2)
    n = len(points)
    parent = -np.ones(n, dtype=np.int64)
    cur = np.zeros(n, dtype=np.int64)
    q_trace, iters, cur_levels = [], [], []
    for l in range(max_level):
        prev_q = 0.0
        it = 0
        while True:
            m0, m1, m2, cur, _ = e_step(points, pi, mu, cov, parent)
            m_step(m0, m1, m2, l, pi, mu, cov, n, ld)
            q = log_likelihood(points, pi, mu, cov, l)
            q_trace.append(q)
            it += 1
            if abs(q - prev_q) < ls or it >= max_iters_per_level:
                break
            prev_q = q
        iters.append(it)
        cur_levels.append(cur.copy())
        parent = cur.copy()
    return pi, mu, cov, BuildTrace(np.array(q_trace), np.array(iters), cur_levels)



def build_flat_fullcov(points, J, ls, ld, init_idx, sig2=0.00034, max_iters=10000):
    """Flat full-covariance EM over J components == ONE tree level with branching J
    (the CPU twin run with its module global ``n_node`` set to J and maxTreeLevel = 1;
    hgmm_cupy_cpu_working.py:30,122-160).  Note pi0 = 1/J and the stop rule on q.

    Returns (pi, mu, cov, q_trace, current_idx)."""
    points = np.asarray(points, dtype=np.float64)
    n = len(points)
    pi = np.full(J, 1.0 / J)
    mu = np.array(points[np.asarray(init_idx)], dtype=np.float64)
    cov = np.tile(np.identity(3) * sig2, (J, 1, 1))
    qs = []
    prev_q = 0.0
    cur = None
    while True:
        ok, inv, coef = node_prep(cov)
        g = pi[None, :] * pdf_pairs(points[:, None, :], mu[None], inv[None], coef[None])
        den = g.sum(axis=1)
        good = den > EPS
        gamma = np.where(good[:, None], g / np.where(good, den, 1.0)[:, None], 0.0)
        cur = np.argmax(gamma, axis=1)
        use = np.where(gamma < EPS, 0.0, gamma)
        m0 = use.sum(axis=0)
        m1 = use.T @ points
        m2 = np.einsum('nj,na,nb->jab', use, points, points)
        for j in range(J):
            if m0[j] < ld:
                pi[j], mu[j], cov[j] = 0.0, 0.0, np.identity(3)
            else:
                pi[j] = m0[j] / n
                mu[j] = m1[j] / m0[j]
                cov[j] = m2[j] / m0[j] - np.outer(mu[j], mu[j])
        sel = ~(pi < EPS)
        ok, inv, coef = node_prep(cov[sel])
        p = pdf_pairs(points[:, None, :], mu[sel][None], inv[None], coef[None])
        q = np.log(np.maximum((p * pi[sel][None, :]).sum(axis=1), EPS)).sum()
        qs.append(q)
        if abs(q - prev_q) < ls or len(qs) >= max_iters:
            break
        prev_q = q
    return pi, mu, cov, np.array(qs), cur


def complexity(cov):
    """smallest eigenvalue / trace (hgmm_cupy_cpu_working.py:87-91; reference uses
    ``np.linalg.eig`` on the symmetric covariance and sorts descending)."""
    lam = np.linalg.eigvalsh(cov)
    return lam[..., 0] / lam.sum(axis=-1)


def reg_e_step(points, pi, mu, cov, max_level, lc):
    """Registration E-step: per point descend the tree; at each level normalise gamma over
    the 8 children of the current node, move to the arg-max child, stop (BEFORE accumulating)
    when its covariance is 'flat enough', otherwise add (g, g x, g x x^T) to that node.
    hgmm_cupy_cpu_working.py:202-228.  Returns (m0[T], m1[T,3], m2[T,3,3])."""
    points = np.asarray(points, dtype=np.float64)
    T = n_total(max_level)
    ok, inv, coef = node_prep(cov)
    cplx = complexity(cov)
    m0 = np.zeros(T)
    m1 = np.zeros((T, 3))
    m2 = np.zeros((T, 3, 3))
    n = len(points)
    search = -np.ones(n, dtype=np.int64)
    alive = np.ones(n, dtype=bool)
    for _ in range(max_level):
        idx = np.nonzero(alive)[0]
        if len(idx) == 0:
            break
        x = points[idx]
        j0 = child(search[idx])
        kid = j0[:, None] + np.arange(N_NODE)[None, :]
        g = pi[kid] * pdf_pairs(x[:, None, :], mu[kid], inv[kid], coef[kid])
        den = g.sum(axis=1)
        good = den > EPS
        gamma = np.where(good[:, None], g / np.where(good, den, 1.0)[:, None], 0.0)
        am = np.argmax(gamma, axis=1)
        s = j0 + am
        search[idx] = s
        stop = cplx[s] <= lc
        alive[idx[stop]] = False
        keep = ~stop
        gs = gamma[np.arange(len(idx)), am][keep]
        gs = np.where(gs < EPS, 0.0, gs)
        sk = s[keep]
        xk = x[keep]
        np.add.at(m0, sk, gs)
        np.add.at(m1, sk, gs[:, None] * xk)
        np.add.at(m2, sk, gs[:, None, None] * (xk[:, :, None] * xk[:, None, :]))
    return m0, m1, m2


# ---------------------------------------------------------------------------
# host-side registration M-step (stays on the host in the product too)
# ---------------------------------------------------------------------------

def skew(x):
    return np.array([[0.0, -x[2], x[1]], [x[2], 0.0, -x[0]], [-x[1], x[0], 0.0]])


def twist_trans(tw):
    """Rodrigues; hgmm_cupy_cpu_working.py:296-314 (non-linear branch)."""
    twd = np.linalg.norm(tw[:3])
    if twd == 0.0:
        return np.identity(3), tw[3:]
    a = tw[:3] / twd
    c, s = np.cos(twd), np.sin(twd)
    return c * np.identity(3) + (1.0 - c) * np.outer(a, a) + s * skew(a), tw[3:]


def twist_mul(tw, rot, t):
    """hgmm_cupy_cpu_working.py:284-294."""
    tr, tt = twist_trans(tw)
    return np.dot(tr, rot), np.dot(t, tr.T) + tt


def reg_m_step(m0, m1, m2, mu, cov, rot, t):
    """Twist least-squares update; hgmm_cupy_cpu_working.py:356-375.
    Returns (rot, t, q) with q = lstsq residual array (may be empty)."""
    n = len(m0)
    amat = np.zeros((n * 3, 6))
    bmat = np.zeros(n * 3)
    for i in range(n):
        if m0[i] < F32_EPS:
            continue
        lam, vec = np.linalg.eigh(cov[i])
        s = m1[i] / m0[i]
        vec = vec * np.sqrt(m0[i] / lam)
        sl = slice(3 * i, 3 * i + 3)
        bmat[sl] = vec.T @ mu[i] - vec.T @ s
        amat[sl, :3] = np.cross(s[None, :], vec.T)
        amat[sl, 3:] = vec.T
    x, q, _, _ = np.linalg.lstsq(amat, bmat, rcond=-1)
    rot, t = twist_mul(x, rot, t)
    return rot, t, q


def register(target, pi, mu, cov, max_level, lc=0.01, maxiter=20, tol=1.0e-4):
    """hgmm_cupy_cpu_working.py:377-391.  Returns (rot_inv, t_inv, q, trace) where
    (rot_inv, t_inv) is ``tf.inverse()`` as the reference returns it."""
    target = np.asarray(target, dtype=np.float64)
    rot, t = np.identity(3), np.zeros(3)
    q_prev = None
    trace = []
    q = None
    for _ in range(maxiter):
        tt = target @ rot.T + t
        m0, m1, m2 = reg_e_step(tt, pi, mu, cov, max_level, lc)
        rot, t, q = reg_m_step(m0, m1, m2, mu, cov, rot, t)
        trace.append((rot.copy(), t.copy(), np.array(q, copy=True), m0, m1, m2))
        if q_prev is not None and q.size and q_prev.size and abs(q - q_prev) < tol:
            break
        q_prev = q
    return rot.T, -rot.T @ t, q, trace
