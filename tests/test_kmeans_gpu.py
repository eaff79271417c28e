"""GPU parity tests of the KMeans initialiser (csrc/kmeans_kernels.hip + hgmm_amd/kmeans.py)
against the outputs of the reference's own ``init_gmm_params`` (tests/golden/kmeans_init.npz,
produced by scikit-learn through the reference), the NumPy oracle and -- where installed --
live scikit-learn.

Arithmetic is float64 on both sides, so seeds (point indices), labels and iteration counts must
be identical and centres agree to summation-order noise (1e-12 absolute on unit-scale data)."""
import time

import numpy as np
import pytest

from conftest import load_golden
from oracle import kmeans as okm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import hgmm_amd
    c = hgmm_amd.Context(0)
    yield c
    c.close()


def _case(g, name, bunny):
    X = bunny[::10].astype(np.float64) if name == "bunny" else g[name + "_X"]
    return X, int(g[name + "_k"])


@pytest.mark.parametrize("name", ["uniform", "blobs", "bunny"])
def test_fit_matches_reference_init(ctx, bunny, name):
    from hgmm_amd.kmeans import KMeans
    g = load_golden("kmeans_init.npz")
    X, k = _case(g, name, bunny)
    km = KMeans(n_clusters=k, random_state=1, max_iter=50, n_init=1, ctx=ctx).fit(X)
    assert np.array_equal(km.init_indices_, g[name + "_init_indices"])
    assert km.n_iter_ == int(g[name + "_n_iter"])
    assert np.array_equal(km.labels_, g[name + "_labels"])
    np.testing.assert_allclose(km.cluster_centers_, g[name + "_centres"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(km.inertia_, float(g[name + "_inertia"]), rtol=1e-12)


def test_relocation_of_empty_clusters(ctx):
    from hgmm_amd.kmeans import KMeans
    g = load_golden("kmeans_init.npz")
    km = KMeans(n_clusters=6, init=g["reloc_init"], max_iter=50, ctx=ctx).fit(g["reloc_X"])
    assert km.n_iter_ == int(g["reloc_n_iter"]) and np.array_equal(km.labels_, g["reloc_labels"])
    np.testing.assert_allclose(km.cluster_centers_, g["reloc_centres"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(km.inertia_, float(g["reloc_inertia"]), rtol=1e-12)


def test_init_gmm_params_drop_in(ctx, bunny):
    """gmmreg_gpu.gmm_impl.init_gmm_params(X, k) -> (means, weights), as the reference returns them."""
    import hgmm_amd
    from hgmm_amd.gmmreg_gpu import gmm_impl
    g = load_golden("kmeans_init.npz")
    hgmm_amd.set_default_context(ctx)
    X, k = _case(g, "bunny", bunny)
    means, weights = gmm_impl.init_gmm_params(X, k)
    np.testing.assert_allclose(means, g["bunny_centres"], rtol=0, atol=1e-12)
    assert np.array_equal(weights, g["bunny_weights"])


@pytest.mark.parametrize("n,k", [(1, 1), (7, 7), (255, 3), (257, 64), (1000, 65), (5000, 300), (20000, 1100)])
def test_seeding_and_step_vs_oracle(ctx, n, k):
    """Ragged sizes, k across the 64-centre slot / 256-centre / 1024-centre pass boundaries:
    seeds identical to the oracle's, one Lloyd step identical in labels and sums."""
    rs = np.random.RandomState(n + k)
    X = rs.rand(n, 3) * [1.0, 2.0, 0.5]
    Xc = X - X.mean(axis=0)
    ctx.set_points(Xc)
    seeds = np.random.RandomState(1)
    trials = okm.n_local_trials(k)
    first = seeds.choice(n, p=np.ones(n) / n)
    rand = seeds.uniform(size=(k - 1, trials))
    ids, centres = ctx.kmeans_plusplus(k, first, rand)
    o_centres, o_ids = okm.kmeans_plusplus(Xc, k, np.random.RandomState(1))
    assert np.array_equal(ids, o_ids)
    assert np.array_equal(centres, o_centres)
    sums, counts, inertia, changed = ctx.kmeans_step(centres, reset_labels=True)
    lab, d2 = okm.assign(Xc, centres)
    labels, dist = ctx.kmeans_labels(with_distances=True)
    assert np.array_equal(labels, lab) and changed == n
    np.testing.assert_allclose(dist, d2, rtol=1e-13, atol=1e-30)
    assert np.array_equal(counts, np.bincount(lab, minlength=k))
    o_sums = np.zeros((k, 3))
    np.add.at(o_sums, lab, Xc)
    np.testing.assert_allclose(sums, o_sums, rtol=0, atol=1e-11)
    np.testing.assert_allclose(inertia, d2.sum(), rtol=1e-12)
    # a second step with the same centres changes nothing
    _, _, _, changed2 = ctx.kmeans_step(centres)
    assert changed2 == 0


@pytest.mark.parametrize("n", [4_300_000, 16_900_000])
def test_seeding_large_clouds_take_the_multi_group_paths(ctx, n):
    """More than 1024 step workgroups (a thread of the tail owns two group sums) and more than 4096 (the group
    prefix leaves LDS for global memory): seeds still the oracle's."""
    k = 5
    rs = np.random.RandomState(n % 1000)
    X = rs.rand(n, 3) * [1.0, 2.0, 0.5]
    Xc = X - X.mean(axis=0)
    del X
    ctx.set_points(Xc)
    seeds = np.random.RandomState(1)
    trials = okm.n_local_trials(k)
    first = seeds.choice(n, p=np.ones(n) / n)
    rand = seeds.uniform(size=(k - 1, trials))
    ids, centres = ctx.kmeans_plusplus(k, first, rand)
    o_centres, o_ids = okm.kmeans_plusplus(Xc, k, np.random.RandomState(1))
    assert np.array_equal(ids, o_ids)
    assert np.array_equal(centres, o_centres)


def test_live_sklearn_agrees(ctx):
    sk = pytest.importorskip("sklearn.cluster")
    from hgmm_amd.kmeans import KMeans
    rs = np.random.RandomState(9)
    X = rs.rand(30000, 3) * [1.0, 0.6, 0.3]
    ref = sk.KMeans(n_clusters=40, random_state=1, max_iter=50, n_init=1).fit(X)
    km = KMeans(n_clusters=40, random_state=1, max_iter=50, n_init=1, ctx=ctx).fit(X)
    assert km.n_iter_ == ref.n_iter_ and np.array_equal(km.labels_, ref.labels_)
    np.testing.assert_allclose(km.cluster_centers_, ref.cluster_centers_, rtol=0, atol=1e-12)


def test_waymo_scale_properties(ctx):
    """BASELINE C3 size (N = 1e6, k = 800): properties that do not need an O(N k) oracle."""
    from hgmm_amd.kmeans import KMeans
    n, k = 1_000_000, 800
    X = np.random.RandomState(0).rand(n, 3)
    t0 = time.perf_counter()
    km = KMeans(n_clusters=k, random_state=1, max_iter=50, n_init=1, ctx=ctx).fit(X)
    dt = time.perf_counter() - t0
    print("KMeans N=1e6 k=800: %.3f s, %d Lloyd iterations, inertia %.6g" % (dt, km.n_iter_, km.inertia_))
    ids = km.init_indices_
    assert len(np.unique(ids)) == k and ids.min() >= 0 and ids.max() < n
    lab = km.labels_
    counts = np.bincount(lab, minlength=k)
    assert counts.sum() == n and counts.min() > 0
    # inertia is the sum of squared distances to the reported centres
    d = ((X - km.cluster_centers_[lab]) ** 2).sum(axis=1)
    np.testing.assert_allclose(km.inertia_, d.sum(), rtol=1e-10)
    # labels are nearest-centre assignments of the final centres (checked on a sample)
    s = np.random.RandomState(1).choice(n, 2000, replace=False)
    mean = X.mean(axis=0)
    o_lab, _ = okm.assign(X[s] - mean, km.cluster_centers_ - mean)
    assert (o_lab == lab[s]).mean() > 0.999          # exact ties aside
    # run-to-run bitwise reproducible
    km2 = KMeans(n_clusters=k, random_state=1, max_iter=50, n_init=1, ctx=ctx).fit(X)
    assert np.array_equal(km2.cluster_centers_, km.cluster_centers_) and np.array_equal(km2.labels_, lab)


@pytest.mark.parametrize("n,k,trials", [(60000, 1200, None), (4000, 50, 16), (4000, 50, 1), (70000, 3, 3),
                                        (300, 2, 2), (1100000, 64, 5)])
def test_seeding_forms_agree(ctx, n, k, trials):
    """The two forms of k-means++ seeding in the library pick the same points, bit for bit: one launch per centre
    (kmpp_fused_kernel: every workgroup of the pass runs the previous centre's tail for itself; both the 8- and the
    16-candidate instantiation) and two launches per centre (option kmpp_two_launches: what clouds beyond 16.7 M points
    take).  (The four-kernel form of round 1 left the library in round 6; the seeds are held to scikit-learn's by
    test_fit_matches_reference_init.)"""
    X = np.random.RandomState(n).rand(n, 3)
    ctx.set_points(X - X.mean(0))
    t = trials or (2 + int(np.log(k)))
    rand = np.random.RandomState(1).uniform(size=(k - 1, t))
    ids, centres = ctx.kmeans_plusplus(k, 7 % n, rand)
    with ctx.config(kmpp_two_launches=1):
        ids2, centres2 = ctx.kmeans_plusplus(k, 7 % n, rand)
    assert np.array_equal(ids, ids2) and np.array_equal(centres, centres2)
    again = ctx.kmeans_plusplus(k, 7 % n, rand)
    assert np.array_equal(again[0], ids) and np.array_equal(again[1], centres)
