#!/usr/bin/env python3
"""Full-size oracle fixtures for the sizes bench.py prints numbers for (task 1).

    python tools/gen_oracle_fixtures.py [--only c4|tree1m|fullcov1m]

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


if __name__ == "__main__":
    main()
