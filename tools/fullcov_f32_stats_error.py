#!/usr/bin/env python3
"""How accurate are full-covariance sufficient statistics whose products gamma x feature are accumulated in FLOAT32 (the
matrix cores' fp32 accumulators of full_fused_f32_kernel) per segment of points, the segments added in float64?
NumPy model on the bench's kind of data (uniform unit cube, spherical mixture sigma = 0.03): features about the first
point or about the centroid, segment lengths 256 ... 1024 points.  CPU only.
    python tools/fullcov_f32_stats_error.py"""
import numpy as np


def cov_from(M):
    m0 = M[:, 0]
    m = M[:, 1:4] / m0[:, None]
    idx = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    c = np.stack([M[:, 4 + t] / m0 - m[:, a] * m[:, b] for t, (a, b) in enumerate(idx)], 1)
    return m, c


def feats(D):
    return np.stack([np.ones(len(D)), D[:, 0], D[:, 1], D[:, 2], D[:, 0] ** 2, D[:, 0] * D[:, 1], D[:, 0] * D[:, 2],
                     D[:, 1] ** 2, D[:, 1] * D[:, 2], D[:, 2] ** 2], 1)


def main():
    rs = np.random.RandomState(0)
    n, J, sig = 262144, 50, 0.03
    X = rs.rand(n, 3)
    mu = rs.rand(J, 3)
    d2 = ((X[:, None, :] - mu[None]) ** 2).sum(-1)
    g = np.exp(-0.5 * d2 / sig ** 2)
    g /= g.sum(1, keepdims=True) + 1e-300
    for name, o in (("first point", X[0]), ("centroid", X.mean(0))):
        D = X - o
        M64 = g.T @ feats(D)
        m64, c64 = cov_from(M64)
        g32, F32 = g.astype(np.float32), feats(D.astype(np.float32).astype(np.float64)).astype(np.float32)
        for seg in (1024, 512, 256):
            T = n // seg
            part = np.zeros((T, J, 10), np.float32)
            for k in range(0, seg, 4):                         # the matrix cores add four points' products per step
                part += np.einsum('tpj,tpf->tjf', g32.reshape(T, seg, J)[:, k:k + 4], F32.reshape(T, seg, 10)[:, k:k + 4],
                                  dtype=np.float32, optimize=False)
            m, c = cov_from(part.astype(np.float64).sum(0))
            print("origin = %-11s segments of %4d points: max |d cov| / sigma^2 = %.2e, max |d mu| = %.2e"
                  % (name, seg, np.abs(c - c64).max() / sig ** 2, np.abs(m - m64).max()))


if __name__ == "__main__":
    main()
