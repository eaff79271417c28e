#!/usr/bin/env python3
"""Full-size oracle fixtures for the sizes bench.py prints numbers for (task 1).

    python tools/gen_oracle_fixtures.py [--only c4|tree1m|fullcov1m|flat1m|flat1m_long|flat2x1m|flat8x1m]

Unlike tools/gen_golden.py (which runs the REFERENCE and needs /root/reference) this script runs the
committed NumPy oracle (oracle/hgmm_tree.py, itself pinned to reference-generated fixtures by
tests/test_oracle_golden.py) at sizes where the oracle needs minutes, so that the GPU tests can
compare against stored results instead of properties:

  c4         BASELINE configs[3] at FULL size: oracle.build_tree on all 40 256 points of bun000, L = 4,
             CPU-twin constants (ls = 80, ld = 1e-4, sig2 = 0.00034, seed-72 initial means)
             -> tests/golden/hgmm_build_bun000_L4_oracle.npz
               (iteration counts, q trace, currentIdx of every level, pi / mu / cov)

  tree1m     the `tree_1M` leg of bench.py: uniform cloud N = 1e6 (RandomState(0), float32 -> float64), L = 4,
             4 iterations per level, sig2 = 0.01 -> tests/golden/hgmm_build_uniform1M_L4_oracle.npz
               (q trace, pi / mu / cov, node populations per level, a position-weighted checksum of currentIdx
                over ALL points per level, currentIdx of 20 000 sampled points)
  fullcov1m  the `fullcov` leg of bench.py: the same cloud, flat full-covariance EM with J = 800, 3 iterations
             (oracle.build_flat_fullcov's op sequence applied to 8192-point chunks: the N x J matrix does not
              fit host memory in one piece) -> tests/golden/fullcov_uniform1M_J800_oracle.npz

  flat1m     the HEADLINE leg of bench.py (BASELINE configs[2]): the same cloud, flat diag / spherical EM with
             J = 800, 3 iterations, all three flavours the reference has, oracle.flat_em's op sequence in float64
             applied to 16384-point chunks (the E-step / M-step split over row blocks; the chunked driver is
             asserted equal to oracle.flat_em.train on a size the oracle takes in one piece)
             -> tests/golden/flat_uniform1M_J800_oracle.npz
               (per flavour: lls[3], mu, w, cov, inv_std after 3 iterations; hard labels of ALL points at the
                float32 roundings of those parameters: population, position-weighted checksum over the rows
                whose two largest responsibilities are at least 1e-5 apart -- north_star's near-tie rule; three
                iterations from cov = 0.1 leave the responsibilities of a uniform cloud flat, 2-3 % of the rows are
                that close to a tie --, those rows themselves with their labels, 20 000 sampled labels + gaps)

Fixtures hold inputs (seeds / constants) and outputs only.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

from oracle import hgmm_tree  # noqa: E402
from oracle import flat_em  # noqa: E402


def gen_c4():
    P = np.load(os.path.join(GOLD, "bun000_xyz.npy")).astype(np.float64)
    L = 4
    T = hgmm_tree.n_total(L)
    idx = np.random.RandomState(72).randint(T, size=T)
    t0 = time.time()
    pi, mu, cov, tr = hgmm_tree.build_tree(P, L, 80.0, 1e-4, idx, 0.00034)
    dt = time.time() - t0
    out = {"L": L, "ls": 80.0, "ld": 1e-4, "sig2": 0.00034, "init_seed": 72, "init_idx": idx.astype(np.int32),
           "n_points": len(P), "iters_per_level": np.asarray(tr.iters_per_level, dtype=np.int32),
           "q_trace": np.asarray(tr.q), "pi": pi, "mu": mu, "cov": cov, "oracle_seconds": dt}
    for l, cur in enumerate(tr.current_idx_per_level):
        out["current_idx_L%d" % l] = np.asarray(cur, dtype=np.int16)      # T = 4680 < 32768
    path = os.path.join(GOLD, "hgmm_build_bun000_L4_oracle.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: iterations %s, q_final %.6f, %.0f s" % (path, list(tr.iters_per_level), tr.q[-1], dt))


def uniform_cloud():
    return np.random.RandomState(0).rand(1_000_000, 3).astype(np.float32).astype(np.float64)    # bench.synth_frame(0)


def label_checksum(cur):
    """Position-weighted checksum of an integer label vector (exact in uint64 arithmetic mod 2^64)."""
    cur = np.asarray(cur).astype(np.uint64)
    pos = np.arange(1, len(cur) + 1, dtype=np.uint64)
    return int((cur * pos).sum(dtype=np.uint64)), int((cur * cur * pos).sum(dtype=np.uint64))


def gen_tree1m():
    P = uniform_cloud()
    L = 4
    T = hgmm_tree.n_total(L)
    idx = np.random.RandomState(72).randint(len(P), size=T)
    t0 = time.time()
    pi, mu, cov, tr = hgmm_tree.build_tree(P, L, 80.0, 1e-4, idx, 0.01, max_iters_per_level=4)
    dt = time.time() - t0
    sample = np.sort(np.random.RandomState(11).choice(len(P), 20000, replace=False))
    out = {"L": L, "ls": 80.0, "ld": 1e-4, "sig2": 0.01, "max_iters_per_level": 4, "cloud_seed": 0, "init_seed": 72,
           "n_points": len(P), "iters_per_level": np.asarray(tr.iters_per_level, dtype=np.int32),
           "q_trace": np.asarray(tr.q), "pi": pi, "mu": mu, "cov": cov, "sample": sample.astype(np.int32),
           "oracle_seconds": dt}
    for l, cur in enumerate(tr.current_idx_per_level):
        out["population_L%d" % l] = np.bincount(cur, minlength=T).astype(np.int32)
        out["checksum_L%d" % l] = np.array(label_checksum(cur), dtype=np.uint64)
        out["current_idx_sample_L%d" % l] = np.asarray(cur[sample], dtype=np.int16)
    path = os.path.join(GOLD, "hgmm_build_uniform1M_L4_oracle.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: iterations %s, q %s, %.0f s" % (path, list(tr.iters_per_level), tr.q, dt))


def fullcov_chunked(P, J, ls, ld, init_idx, sig2, max_iters, chunk=8192):
    """oracle.hgmm_tree.build_flat_fullcov (hgmm_cupy_cpu_working.py:122-198 with n_node = J) with the N x J
    responsibility matrix formed chunk by chunk; same primitives (node_prep, pdf_pairs), same rules."""
    EPS = hgmm_tree.EPS
    n = len(P)
    pi = np.full(J, 1.0 / J)
    mu = np.array(P[np.asarray(init_idx)], dtype=np.float64)
    cov = np.tile(np.identity(3) * sig2, (J, 1, 1))
    qs, prev_q = [], 0.0
    cur = np.zeros(n, dtype=np.int64)
    while True:
        ok, inv, coef = hgmm_tree.node_prep(cov)
        m0, m1, m2 = np.zeros(J), np.zeros((J, 3)), np.zeros((J, 3, 3))
        for s in range(0, n, chunk):
            x = P[s:s + chunk]
            g = pi[None, :] * hgmm_tree.pdf_pairs(x[:, None, :], mu[None], inv[None], coef[None])
            den = g.sum(axis=1)
            good = den > EPS
            gamma = np.where(good[:, None], g / np.where(good, den, 1.0)[:, None], 0.0)
            cur[s:s + chunk] = np.argmax(gamma, axis=1)
            use = np.where(gamma < EPS, 0.0, gamma)
            m0 += use.sum(axis=0)
            m1 += use.T @ x
            m2 += np.einsum('nj,na,nb->jab', use, x, x)
        for j in range(J):
            if m0[j] < ld:
                pi[j], mu[j], cov[j] = 0.0, 0.0, np.identity(3)
            else:
                pi[j] = m0[j] / n
                mu[j] = m1[j] / m0[j]
                cov[j] = m2[j] / m0[j] - np.outer(mu[j], mu[j])
        sel = ~(pi < EPS)
        ok, inv, coef = hgmm_tree.node_prep(cov[sel])
        q = 0.0
        for s in range(0, n, chunk):
            x = P[s:s + chunk]
            p = hgmm_tree.pdf_pairs(x[:, None, :], mu[sel][None], inv[None], coef[None])
            q += np.log(np.maximum((p * pi[sel][None, :]).sum(axis=1), EPS)).sum()
        qs.append(q)
        if abs(q - prev_q) < ls or len(qs) >= max_iters:
            break
        prev_q = q
    return pi, mu, cov, np.array(qs), cur


def gen_fullcov1m():
    # the chunked driver == the oracle on a size the oracle takes in one piece
    Ps = uniform_cloud()[:3000]
    ii = np.random.RandomState(3).choice(len(Ps), 40, replace=False)
    a = hgmm_tree.build_flat_fullcov(Ps, 40, 1e-30, 1e-4, ii, 0.01, max_iters=4)
    b = fullcov_chunked(Ps, 40, 1e-30, 1e-4, ii, 0.01, 4, chunk=512)
    assert np.array_equal(a[4], b[4]) and np.allclose(a[3], b[3], rtol=1e-12) and np.allclose(a[2], b[2], rtol=1e-9, atol=1e-16)
    P = uniform_cloud()
    J = 800
    idx = np.random.RandomState(100).choice(len(P), J, replace=False)
    t0 = time.time()
    pi, mu, cov, qs, cur = fullcov_chunked(P, J, 1e-30, 1e-4, idx, 0.01, 3)
    dt = time.time() - t0
    sample = np.sort(np.random.RandomState(12).choice(len(P), 20000, replace=False))
    out = {"J": J, "ls": 1e-30, "ld": 1e-4, "sig2": 0.01, "max_iters": 3, "cloud_seed": 0, "init_seed": 100,
           "n_points": len(P), "q_trace": qs, "pi": pi, "mu": mu, "cov": cov, "sample": sample.astype(np.int32),
           "labels_sample": cur[sample].astype(np.int16), "population": np.bincount(cur, minlength=J).astype(np.int32),
           "checksum": np.array(label_checksum(cur), dtype=np.uint64), "oracle_seconds": dt}
    path = os.path.join(GOLD, "fullcov_uniform1M_J800_oracle.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: q %s, %.0f s" % (path, qs, dt))


def flat_m_from_sums(n, s0, s1, s2, cov_type, variant):
    """oracle.flat_em.m_step (gmm_waymo gmm_impl.py:81-103 / gmmreg_gpu gmm_impl.py:46-52) with the three products
    resp.T @ 1, resp.T @ X, resp.T @ (X * X) handed in as sums over row blocks."""
    EPS = flat_em.EPS
    if variant == "W":
        nk = s0 + EPS
        mu = s1 / nk[:, None]
        ex2 = s2 / nk[:, None]
        cov = ex2 - 2 * (mu * s1 / nk[:, None]) + mu ** 2 + flat_em.REG_COVAR
        if cov_type == "spherical":
            cov = np.mean(cov, axis=1)
        return nk / n, mu, cov
    mu = s1 / (s0[:, None] + EPS)
    ex2 = s2 / (s0[:, None] + EPS)
    return s0 / n, mu, np.clip(ex2 - mu ** 2, 0.0, None)


def flat_chunked(X, iters, mu, cov, w, cov_type, variant, chunk=16384):
    """oracle.flat_em.train (gmm_impl.py:118-145 / 63-85; tol = 0) with the N x J tables formed per row block."""
    n, J = len(X), len(mu)
    inv = flat_em.inv_std_from_cov(cov, variant, initial=True)
    lls = []
    for _ in range(iters):
        s0, s1, s2, lsum = np.zeros(J), np.zeros((J, 3)), np.zeros((J, 3)), 0.0
        for s in range(0, n, chunk):
            x = X[s:s + chunk]
            _, log_resp, lpn, _ = flat_em.e_step_full(x, inv, mu, w, cov_type, variant)
            r = np.exp(log_resp)
            lsum += lpn.sum()
            s0 += r.sum(axis=0)
            s1 += r.T @ x
            s2 += r.T @ (x * x)
        lls.append(lsum / n)
        w, mu, cov = flat_m_from_sums(n, s0, s1, s2, cov_type, variant)
        inv = flat_em.inv_std_from_cov(cov, variant)
    return inv, mu, w, cov, np.array(lls)


def flat_labels_chunked(X, inv, mu, w, cov_type, variant, chunk=16384):
    """oracle.flat_em.predict over row blocks + the gap between the two largest responsibilities of every row."""
    lab = np.zeros(len(X), dtype=np.int64)
    gap = np.zeros(len(X))
    for s in range(0, len(X), chunk):
        _, log_resp, _, am = flat_em.e_step_full(X[s:s + chunk], inv, mu, w, cov_type, variant)
        part = np.partition(np.exp(log_resp), -2, axis=1)
        lab[s:s + chunk] = am
        gap[s:s + chunk] = part[:, -1] - part[:, -2]
    return lab, gap


FLAT_FLAVOURS = (("W", "diag"), ("W", "spherical"), ("G", "diag"))


def gen_flat1m():
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    # the chunked driver == the oracle on a size the oracle takes in one piece
    Xs = f64(np.random.RandomState(0).rand(5000, 3).astype(np.float32))
    for variant, cov_type in FLAT_FLAVOURS:
        mu0, w0, cov0 = flat_em.seeded_init(Xs.astype(np.float32), 40, 3, cov_type)
        a = flat_em.train(Xs, 4, 0.0, f64(mu0), f64(cov0), f64(w0), cov_type, variant)
        b = flat_chunked(Xs, 4, f64(mu0), f64(cov0), f64(w0), cov_type, variant, chunk=512)
        assert np.allclose(a[4], b[4], rtol=1e-13, atol=0) and np.allclose(a[1], b[1], rtol=1e-11, atol=1e-14)
        assert np.allclose(a[2], b[2], rtol=1e-11) and np.allclose(a[3], b[3], rtol=1e-9, atol=1e-15)
        assert np.array_equal(flat_labels_chunked(Xs, a[0], a[1], a[2], cov_type, variant, 512)[0],
                              flat_em.predict(Xs, a[0], a[1], a[2], cov_type, variant))
    N, J, iters = 1_000_000, 800, 3
    X32 = np.random.RandomState(0).rand(N, 3).astype(np.float32)            # bench.synth_frame(0)
    X = f64(X32)
    idx = np.random.RandomState(100).choice(N, J, replace=False)            # bench.synth_init: seed 100 + frame
    sample = np.sort(np.random.RandomState(13).choice(N, 20000, replace=False))
    out = {"N": N, "J": J, "iters": iters, "cloud_seed": 0, "init_seed": 100, "init_idx": idx.astype(np.int32),
           "cov0": np.float32(0.1), "sample": sample.astype(np.int32), "tie_gap": 1e-5}
    t00 = time.time()
    for variant, cov_type in FLAT_FLAVOURS:
        t0 = time.time()
        mu0 = f64(X32[idx])
        w0 = f64((np.ones(J) / J).astype(np.float32))
        cov0 = f64((0.1 * np.ones((J, 3) if cov_type == "diag" else (J,))).astype(np.float32))
        inv, mu, w, cov, lls = flat_chunked(X, iters, mu0, cov0, w0, cov_type, variant)
        # hard labels at the float32 roundings of the final parameters: the inputs a float32 engine can be given exactly
        inv32, mu32, w32 = (np.asarray(a, dtype=np.float32) for a in (inv, mu, w))
        lab, gap = flat_labels_chunked(X, f64(inv32), f64(mu32), f64(w32), cov_type, variant)
        near = np.flatnonzero(gap < out["tie_gap"])
        clear = np.ones(N, dtype=bool)
        clear[near] = False
        k = "%s_%s_" % (variant, cov_type)
        out.update({k + "lls": lls, k + "mu": mu, k + "w": w, k + "cov": cov, k + "inv": inv,
                    k + "population": np.bincount(lab, minlength=J).astype(np.int32),
                    k + "checksum_clear": np.array(label_checksum(np.where(clear, lab, 0)), dtype=np.uint64),
                    k + "near_rows": near.astype(np.int32), k + "near_labels": lab[near].astype(np.int16),
                    k + "labels_sample": lab[sample].astype(np.int16),
                    k + "gap_sample": gap[sample].astype(np.float32)})
        print("flat1m %s/%s: lls %s, %d rows within %.0e of a tie, %.0f s"
              % (variant, cov_type, lls, len(near), out["tie_gap"], time.time() - t0), flush=True)
    out["oracle_seconds"] = time.time() - t00
    path = os.path.join(GOLD, "flat_uniform1M_J800_oracle.npz")
    np.savez_compressed(path, **out)
    print("wrote %s, %.0f s" % (path, out["oracle_seconds"]))


def gen_flat1m_long():
    """The same frame and initial parameters, flavour W / diag, 20 iterations: where the fit has left its initial
    parameters far behind (the trajectory the timed region of bench.py runs through)
    -> tests/golden/flat_uniform1M_J800_oracle_20it.npz (lls[20], mu, w, cov, inv_std)."""
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    N, J, iters = 1_000_000, 800, 20
    X32 = np.random.RandomState(0).rand(N, 3).astype(np.float32)
    idx = np.random.RandomState(100).choice(N, J, replace=False)
    t0 = time.time()
    inv, mu, w, cov, lls = flat_chunked(f64(X32), iters, f64(X32[idx]), f64((0.1 * np.ones((J, 3))).astype(np.float32)),
                                        f64((np.ones(J) / J).astype(np.float32)), "diag", "W")
    path = os.path.join(GOLD, "flat_uniform1M_J800_oracle_20it.npz")
    np.savez_compressed(path, N=N, J=J, iters=iters, cloud_seed=0, init_seed=100, init_idx=idx.astype(np.int32),
                        lls=lls, mu=mu, w=w, cov=cov, inv=inv, oracle_seconds=time.time() - t0)
    print("wrote %s: lls %s ... %s, %.0f s" % (path, lls[:3], lls[-3:], time.time() - t0))


def gen_flat2x1m(nframes=2):
    """BASELINE configs[4]: `nframes` of its frames (bench.synth_frame(r): 1M uniform points each, seed = rank) fitted
    jointly, initial parameters from frame 0 as bench.py takes them, flavour W / diag, 3 iterations
    -> tests/golden/flat_uniform<nframes>x1M_J800_oracle.npz: what `nframes` ranks with one frame each must reproduce
    (2 frames: configs[4] in small; 8 frames: configs[4] itself, the whole joint fit of the 8-GPU run)."""
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    N, J, iters = 1_000_000, 800, 3
    frames = [np.random.RandomState(r).rand(N, 3).astype(np.float32) for r in range(nframes)]
    idx = np.random.RandomState(100).choice(N, J, replace=False)
    t0 = time.time()
    inv, mu, w, cov, lls = flat_chunked(f64(np.concatenate(frames)), iters, f64(frames[0][idx]),
                                        f64((0.1 * np.ones((J, 3))).astype(np.float32)),
                                        f64((np.ones(J) / J).astype(np.float32)), "diag", "W")
    path = os.path.join(GOLD, "flat_uniform%dx1M_J800_oracle.npz" % nframes)
    np.savez_compressed(path, N_per_frame=N, frames=nframes, J=J, iters=iters, init_seed=100, init_idx=idx.astype(np.int32),
                        lls=lls, mu=mu, w=w, cov=cov, inv=inv, oracle_seconds=time.time() - t0)
    print("wrote %s: lls %s, %.0f s" % (path, lls, time.time() - t0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    if a.only in ("", "c4"):
        gen_c4()
    if a.only in ("", "tree1m"):
        gen_tree1m()
    if a.only in ("", "fullcov1m"):
        gen_fullcov1m()
    if a.only in ("", "flat1m"):
        gen_flat1m()
    if a.only in ("", "flat1m_long"):
        gen_flat1m_long()
    if a.only in ("", "flat2x1m"):
        gen_flat2x1m(2)
    if a.only in ("", "flat8x1m"):
        gen_flat2x1m(8)


if __name__ == "__main__":
    main()
