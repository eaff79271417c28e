"""Times the device KMeans initialiser (seeding, Lloyd assignment / accumulation kernels, whole
fit) and scikit-learn's KMeans on the same input.  python tools/kmbench.py [--sklearn-big]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hgmm_amd  # noqa: E402
from hgmm_amd.kmeans import KMeans  # noqa: E402


def device_case(ctx, X, k, label):
    n = len(X)
    Xc = X - X.mean(axis=0)
    ctx.set_points(Xc)
    rs = np.random.RandomState(1)
    trials = 2 + int(np.log(k))
    first = rs.choice(n, p=np.ones(n) / n)
    rand = rs.uniform(size=(k - 1, trials))
    ctx.kmeans_plusplus(k, first, rand)
    t0 = time.perf_counter()
    ids, centres = ctx.kmeans_plusplus(k, first, rand)
    t_seed = time.perf_counter() - t0
    ctx.kmeans_step(centres, reset_labels=True)
    ctx.profile_reset()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.kmeans_step(centres)
    t_step = (time.perf_counter() - t0) / 10
    ctx.profile_enable(False)
    a_ms, a_n = ctx.profile_get("kmeans_assign")
    c_ms, c_n = ctx.profile_get("kmeans_accum")
    fits = []
    for _ in range(3):
        t0 = time.perf_counter()
        km = KMeans(n_clusters=k, random_state=1, max_iter=50, n_init=1, ctx=ctx).fit(X)
        fits.append(time.perf_counter() - t0)
    t_fit = min(fits)
    pairs = n * k
    print("%s: N=%d k=%d | seeding %.2f ms (%.1f us/centre) | Lloyd step %.3f ms (assign %.3f ms = %.2e pairs/s, "
          "accumulate %.3f ms) | fit %.1f ms, %d iterations" %
          (label, n, k, t_seed * 1e3, t_seed * 1e6 / k, t_step * 1e3, a_ms / a_n, pairs / (a_ms / a_n * 1e-3),
           c_ms / c_n, t_fit * 1e3, km.n_iter_))
    return km, t_fit


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sklearn-big", action="store_true")
    args = ap.parse_args()
    ctx = hgmm_amd.Context(0)
    bun = np.load(os.path.join(ROOT, "tests", "golden", "bun000_xyz.npy")).astype(np.float64)
    from sklearn.cluster import KMeans as SK
    for X, k, label, run_sk in ((bun, 100, "bun000", True), (bun, 800, "bun000", True),
                                (np.random.RandomState(0).rand(1_000_000, 3), 800, "C3", args.sklearn_big)):
        km, t_fit = device_case(ctx, X, k, label)
        if run_sk:
            t0 = time.perf_counter()
            ref = SK(n_clusters=k, random_state=1, max_iter=50, n_init=1).fit(X)
            t_sk = time.perf_counter() - t0
            same = np.array_equal(ref.labels_, km.labels_)
            print("    scikit-learn (%d host cores): %.1f ms, %d iterations -> x%.0f; labels identical: %s, "
                  "max |centre diff| %.1e" % (os.cpu_count(), t_sk * 1e3, ref.n_iter_, t_sk / t_fit, same,
                                            np.abs(ref.cluster_centers_ - km.cluster_centers_).max()))


if __name__ == "__main__":
    main()
