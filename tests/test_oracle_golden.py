"""Pins the CPU oracle (oracle/*.py) to outputs of the reference itself.

The fixtures under tests/golden/ were produced by tools/gen_golden.py, which imports /
exec's the reference's own Python in the build container.  These tests run anywhere
(no GPU, no /root/reference)."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import flat_em, hgmm_tree

FLAT_CASES = [("W", "diag"), ("W", "spherical"), ("G", "diag")]


@pytest.mark.parametrize("variant,cov_type", FLAT_CASES)
def test_flat_small_matches_reference_bitwise(variant, cov_type):
    g = load_golden("flat_small_%s_%s.npz" % (variant, cov_type))
    X, mu0, w0, cov0 = g["X"], g["mu0"], g["w0"], g["cov0"]
    inv0 = flat_em.inv_std_from_cov(cov0, variant, initial=True)
    ll, lr = flat_em.e_step(X, inv0, mu0, w0, cov_type, variant)
    # same fp32 op sequence as the reference => identical bits
    assert np.array_equal(lr, g["e0_log_resp"])
    assert np.float32(ll) == g["e0_ll"]
    w, mu, cov = flat_em.m_step(X, np.exp(lr), cov_type, variant)
    assert np.array_equal(w, g["m0_w"]) and np.array_equal(mu, g["m0_mu"])
    assert np.array_equal(cov, g["m0_cov"])
    for iters in (1, 5):
        inv, mu, w, cov, lls, _ = flat_em.train(X, iters, 0.0, mu0.copy(), cov0.copy(), w0.copy(),
                                                cov_type, variant)
        pre = "it%d_" % iters
        assert np.array_equal(inv, g[pre + "inv"])
        assert np.array_equal(mu, g[pre + "mu"])
        assert np.array_equal(w, g[pre + "w"])
        assert np.array_equal(cov, g[pre + "cov"])
        assert np.array_equal(np.array(lls, dtype=np.float32), g[pre + "lls"])
        _, lr2 = flat_em.e_step(X, inv, mu, w, cov_type, variant)
        assert np.array_equal(lr2, g[pre + "log_resp"])
        assert np.array_equal(flat_em.predict(X, inv, mu, w, cov_type, variant), g[pre + "predict"])


@pytest.mark.parametrize("variant,cov_type", [("W", "diag"), ("W", "spherical"), ("G", "diag")])
@pytest.mark.parametrize("J", [100, 800])
def test_flat_bunny_matches_reference(bunny, J, variant, cov_type):
    """bun000.ply, seeded init, 20 iterations, tol=0 (BASELINE configs 1/2), every flavour of the reference."""
    name = "flat_bunny_J%d.npz" % J if (variant, cov_type) == ("W", "diag") else \
        "flat_bunny_J%d_%s_%s.npz" % (J, variant, cov_type)
    g = load_golden(name)
    X = bunny
    mu0 = X[g["init_idx"]].copy()
    w0 = (np.ones(J) / J).astype(np.float32)
    cov0 = (0.1 * np.ones((J, 3) if cov_type == "diag" else (J,))).astype(np.float32)
    iters = 20 if (J == 100 and variant == "W" and cov_type == "diag") else 3   # keep the CPU suite short: prefixes
    inv, mu, w, cov, lls, _ = flat_em.train(X, iters, 0.0, mu0, cov0, w0, cov_type, variant)
    # BLAS threading may change GEMM summation order => allow fp32 round-off, not more
    np.testing.assert_allclose(np.array(lls, dtype=np.float32), g["lls"][:iters], rtol=2e-5, atol=2e-5)
    if iters == 20:
        np.testing.assert_allclose(mu, g["mu"], rtol=0, atol=2e-4)
        np.testing.assert_allclose(w, g["w"], rtol=0, atol=1e-5)
    rows = g["rows"]
    inv0 = flat_em.inv_std_from_cov(cov0, variant, initial=True)
    for tag, (a, b, c) in {"init": (inv0, X[g["init_idx"]], w0),
                           "final": (g["inv"], g["mu"], g["w"])}.items():
        _, lr = flat_em.e_step(X[rows].astype(np.float64), a.astype(np.float64),
                               b.astype(np.float64), c.astype(np.float64), cov_type, variant)
        np.testing.assert_allclose(np.exp(lr), g[tag + "_resp64_rows"], rtol=0, atol=1e-12)


# ---------------------------------------------------------------------------
# HGMM (CPU twin) goldens
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["hgmm_build_L2.npz", "hgmm_build_L3.npz"])
def test_hgmm_build_matches_reference(name):
    g = load_golden(name)
    P, L = g["points"], int(g["L"])
    pi, mu, cov, tr = hgmm_tree.build_tree(P, L, float(g["ls"]), float(g["ld"]), g["init_idx"],
                                           float(g["sig2"]))
    assert list(tr.iters_per_level) == list(g["iters_per_level"])
    np.testing.assert_allclose(tr.q, g["q_trace"], rtol=1e-9, atol=1e-6)
    for l in range(L):
        assert np.array_equal(tr.current_idx_per_level[l], g["current_idx_L%d" % l])
    np.testing.assert_allclose(pi, g["pi"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(mu, g["mu"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(cov, g["cov"], rtol=1e-7, atol=1e-13)


def test_hgmm_registration_matches_reference():
    g = load_golden("hgmm_reg_L2.npz")
    pi, mu, cov, L, lc = g["pi"], g["mu"], g["cov"], int(g["L"]), float(g["lambda_c"])
    for deg in (10, 30):
        tag = "rot%d_" % deg
        target = g[tag + "target"]
        m0, m1, m2 = hgmm_tree.reg_e_step(target, pi, mu, cov, L, lc)
        np.testing.assert_allclose(m0, g[tag + "m0"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(m1, g[tag + "m1"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(m2, g[tag + "m2"], rtol=1e-9, atol=1e-12)
        rot, t, q, trace = hgmm_tree.register(target, pi, mu, cov, L, lc, maxiter=5, tol=1e-4)
        # per-iteration callback transforms are tf.inverse()
        for k, (r_k, t_k, _, _, _, _) in enumerate(trace):
            np.testing.assert_allclose(r_k.T, g[tag + "iter_rot"][k], rtol=0, atol=1e-8)
            np.testing.assert_allclose(-r_k.T @ t_k, g[tag + "iter_t"][k], rtol=0, atol=1e-8)
        np.testing.assert_allclose(rot, g[tag + "final_rot"], rtol=0, atol=1e-8)
        np.testing.assert_allclose(t, g[tag + "final_t"], rtol=0, atol=1e-8)
        np.testing.assert_allclose(q, g[tag + "final_q"], rtol=1e-6)


def test_fullcov_flat_matches_reference():
    """Flat full-covariance EM == one tree level with branching J."""
    g = load_golden("fullcov_flat.npz")
    P = g["points"]
    for J in (8, 32):
        tag = "J%d_" % J
        pi, mu, cov, q, cur = hgmm_tree.build_flat_fullcov(P, J, 80.0, 1e-4, g[tag + "init_idx"])
        np.testing.assert_allclose(q, g[tag + "q_trace"], rtol=1e-9, atol=1e-6)
        assert np.array_equal(cur, g[tag + "current_idx"])
        np.testing.assert_allclose(pi, g[tag + "pi"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(mu, g[tag + "mu"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(cov, g[tag + "cov"], rtol=1e-7, atol=1e-13)


# ---------------------------------------------------------------------------------------------
# KMeans initialiser (gmmreg_gpu/gmm_impl.py:18-24 -> scikit-learn)
# ---------------------------------------------------------------------------------------------
def _kmeans_case(g, name, bunny):
    X = bunny[::10].astype(np.float64) if name == "bunny" else g[name + "_X"]
    return X, int(g[name + "_k"])


@pytest.mark.parametrize("name", ["uniform", "blobs", "bunny"])
def test_kmeans_oracle_matches_reference_init(name, bunny):
    """oracle/kmeans.py against the outputs of the reference's own init_gmm_params (same seeds,
    same labels, same iteration count; centres to thread-order noise)."""
    from oracle import kmeans as okm
    g = load_golden("kmeans_init.npz")
    X, k = _kmeans_case(g, name, bunny)
    o = okm.fit(X, k, random_state=1, max_iter=50)
    assert np.array_equal(o["init_indices"], g[name + "_init_indices"])
    assert o["n_iter_"] == int(g[name + "_n_iter"])
    assert np.array_equal(o["labels_"], g[name + "_labels"])
    np.testing.assert_allclose(o["cluster_centers_"], g[name + "_centres"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(o["inertia_"], float(g[name + "_inertia"]), rtol=1e-12)
    assert np.array_equal(g[name + "_weights"], np.ones(k) / k)


def test_kmeans_oracle_relocates_empty_clusters_like_sklearn():
    from oracle import kmeans as okm
    g = load_golden("kmeans_init.npz")
    X, init = g["reloc_X"], g["reloc_init"]
    mean = X.mean(axis=0)
    labels, inertia, centres, n_iter = okm.lloyd(X - mean, init - mean, 50, float(np.mean(np.var(X, axis=0)) * 1e-4))
    assert n_iter == int(g["reloc_n_iter"]) and np.array_equal(labels, g["reloc_labels"])
    np.testing.assert_allclose(centres + mean, g["reloc_centres"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(inertia, float(g["reloc_inertia"]), rtol=1e-12)


def test_kmeans_oracle_matches_live_sklearn():
    """Same check against whatever scikit-learn is installed where the tests run."""
    sk = pytest.importorskip("sklearn.cluster")
    from oracle import kmeans as okm
    rs = np.random.RandomState(5)
    X = rs.rand(1500, 3) * [1.0, 0.5, 0.2]
    km = sk.KMeans(n_clusters=12, random_state=1, max_iter=50, n_init=1).fit(X)
    o = okm.fit(X, 12)
    assert o["n_iter_"] == km.n_iter_ and np.array_equal(o["labels_"], km.labels_)
    np.testing.assert_allclose(o["cluster_centers_"], km.cluster_centers_, rtol=0, atol=1e-12)
