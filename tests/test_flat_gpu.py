"""GPU parity tests of the flat GMM EM path (HIP kernels via the C ABI) against the CPU oracle.

Bar (BASELINE.json north_star): responsibilities within 1e-5 of the CPU EM, identical hard
assignments (except genuine near-ties: top-2 responsibilities closer than 1e-5), parameters
after training within the tolerances written below.  The oracle is evaluated in float64 on the
same float32 inputs; the reference's own float32 noise floor on these inputs is 7e-5..3.5e-4
(BASELINE.md section 2), i.e. the kernels are closer to exact arithmetic than the reference is.
"""
import numpy as np
import pytest

from conftest import load_golden
from oracle import flat_em

pytestmark = pytest.mark.gpu

RESP_TOL = 1e-5


@pytest.fixture(scope="module")
def ctx():
    import hgmm_amd
    c = hgmm_amd.Context(0)
    yield c
    c.close()


def oracle64_estep(X, inv, mu, w, cov_type, variant):
    f = lambda a: np.asarray(a, dtype=np.float64)
    return flat_em.e_step_full(f(X), f(inv), f(mu), f(w), cov_type, variant)


def check_estep(ctx, X, inv, mu, w, cov_type, variant, label=""):
    ctx.set_points(X)
    o_mean, o_lr, o_lpn, o_am = oracle64_estep(X, inv, mu, w, cov_type, variant)
    # Without the arg-max output the materialising kernel takes its constant-shift loop (no row maximum), with it
    # the row-maximum loop: BOTH are held to the oracle, the first here, the second below.
    mean_cs, lr_cs, lpn_cs, _ = ctx.flat_estep(inv, mu, w, cov_type, variant, want_lpn=True, want_argmax=False)
    lr_cs, lpn_cs = lr_cs.get(), lpn_cs.get()
    assert np.abs(np.exp(lr_cs.astype(np.float64)) - np.exp(o_lr)).max() <= RESP_TOL
    assert np.abs(lpn_cs - o_lpn).max() <= 2e-5 * max(1.0, np.abs(o_lpn).max())
    assert abs(mean_cs - o_mean) <= 1e-5 * max(1.0, abs(o_mean))
    np.testing.assert_allclose(lr_cs[o_lr > -30], o_lr[o_lr > -30], rtol=2e-5, atol=2e-5)
    mean, lr, lpn, am = ctx.flat_estep(inv, mu, w, cov_type, variant, want_lpn=True, want_argmax=True)
    lr, lpn, am = lr.get(), lpn.get(), am.get()
    r, o_r = np.exp(lr.astype(np.float64)), np.exp(o_lr)
    d_resp = np.abs(r - o_r).max()
    d_lpn = np.abs(lpn - o_lpn).max()
    print("%s max|dresp|=%.3g max|dlpn|=%.3g mean_lpn %.8f vs %.8f" % (label, d_resp, d_lpn, mean, o_mean))
    assert d_resp <= RESP_TOL
    assert d_lpn <= 2e-5 * max(1.0, np.abs(o_lpn).max())
    assert abs(mean - o_mean) <= 1e-5 * max(1.0, abs(o_mean))
    # log_resp itself: relative on the entries that matter, absolute elsewhere
    big = o_lr > -30
    np.testing.assert_allclose(lr[big], o_lr[big], rtol=2e-5, atol=2e-5)
    # hard assignments: identical except genuine near-ties
    flips = am != o_am
    if flips.any():
        part = np.partition(o_r[flips], -2, axis=1)
        assert ((part[:, -1] - part[:, -2]) < RESP_TOL).all(), "label flip that is not a near-tie"
    print("%s label flips (near-ties) %d / %d" % (label, int(flips.sum()), len(am)))
    return lr


@pytest.mark.parametrize("variant,cov_type", [("W", "diag"), ("W", "spherical"), ("G", "diag")])
def test_estep_small_golden(ctx, variant, cov_type):
    g = load_golden("flat_small_%s_%s.npz" % (variant, cov_type))
    X = g["X"]
    inv0 = flat_em.inv_std_from_cov(g["cov0"], variant, initial=True)
    lr = check_estep(ctx, X, inv0, g["mu0"], g["w0"], cov_type, variant, "small/init")
    # against the reference's own float32 output (its noise floor is far below 1e-5 here)
    assert np.abs(np.exp(lr) - np.exp(g["e0_log_resp"])).max() <= RESP_TOL
    check_estep(ctx, X, g["it5_inv"], g["it5_mu"], g["it5_w"], cov_type, variant, "small/it5")
    ctx.set_points(X)
    lab = ctx.flat_predict(g["it5_inv"], g["it5_mu"], g["it5_w"], cov_type, variant).get()
    assert (lab != g["it5_predict"]).sum() <= 1


FLAVOURS = [("W", "diag"), ("W", "spherical"), ("G", "diag")]


def bunny_golden(J, variant, cov_type):
    name = "flat_bunny_J%d.npz" % J if (variant, cov_type) == ("W", "diag") else \
        "flat_bunny_J%d_%s_%s.npz" % (J, variant, cov_type)
    return load_golden(name)


@pytest.mark.parametrize("variant,cov_type", FLAVOURS)
@pytest.mark.parametrize("J", [100, 800])
def test_estep_bunny(ctx, bunny, J, variant, cov_type):
    """BASELINE configs 1/2: bun000.ply, J=100 / J=800, at the initial and at the 20-iteration
    parameters of the reference run -- every flavour the reference has (gmm_waymo diag / spherical,
    gmmreg_gpu diag)."""
    g = bunny_golden(J, variant, cov_type)
    X = bunny
    w0 = (np.ones(J) / J).astype(np.float32)
    inv0 = (1 / np.sqrt(0.1 * np.ones((J, 3) if cov_type == "diag" else (J,)))).astype(np.float32)
    for tag, (inv, mu, w) in {"init": (inv0, X[g["init_idx"]], w0),
                              "final": (g["inv"], g["mu"], g["w"])}.items():
        lr = check_estep(ctx, X, inv, mu, w, cov_type, variant, "bunny J=%d %s/%s %s" % (J, variant, cov_type, tag))
        rows = g["rows"]
        d64 = np.abs(np.exp(lr[rows].astype(np.float64)) - g[tag + "_resp64_rows"]).max()
        d32 = np.abs(np.exp(lr[rows]) - g[tag + "_resp32_rows"]).max()
        noise = float(g[tag + "_noise_max_abs_dresp"])
        print("   vs reference fp64 rows %.3g ; vs reference fp32 rows %.3g (reference fp32-vs-fp64 noise %.3g)"
              % (d64, d32, noise))
        assert d64 <= RESP_TOL
        # against the reference's own float32 output: no further away than that output is from the reference's
        # float64 run (its noise floor, stored with the fixture) plus the 1e-5 bar
        assert d32 <= noise + RESP_TOL
        # hard assignments vs the reference's float64 run
        ctx.set_points(X)
        lab = ctx.flat_predict(inv, mu, w, cov_type, variant).get()
        flips = lab != g[tag + "_argmax64"].astype(np.int64)
        assert (g[tag + "_top2gap64"][flips] < RESP_TOL).all()
        # ... and vs its float32 run: flips only where the float32 reference itself is within its noise of a tie
        flips32 = lab != g[tag + "_argmax32"].astype(np.int64)
        assert (g[tag + "_top2gap64"][flips32] <= 2 * noise + RESP_TOL).all()


@pytest.mark.parametrize("variant,cov_type", [("W", "diag"), ("W", "spherical"), ("G", "diag")])
def test_mstep_from_resp(ctx, variant, cov_type):
    g = load_golden("flat_small_%s_%s.npz" % (variant, cov_type))
    X = g["X"]
    resp = np.exp(g["e0_log_resp"])
    o_w, o_mu, o_cov = flat_em.m_step(X.astype(np.float64), resp.astype(np.float64), cov_type, variant)
    ctx.set_points(X)
    for hint in (None, g["mu0"]):
        w, mu, cov = ctx.flat_mstep(resp, cov_type, variant, centre_hint=hint)
        np.testing.assert_allclose(w, o_w, rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(mu, o_mu, rtol=0, atol=2e-6)
        np.testing.assert_allclose(cov, o_cov, rtol=1e-4 if hint is None else 2e-5, atol=1e-9)
    # fused exp: m_step(X, log_resp.exp()) == m_step(X, exp(log_resp))
    lr_dev = ctx.to_device(g["e0_log_resp"])
    w2, mu2, cov2 = ctx.flat_mstep(lr_dev.exp(), cov_type, variant, centre_hint=g["mu0"])
    np.testing.assert_allclose(mu2, o_mu, rtol=0, atol=2e-6)
    np.testing.assert_allclose(cov2, o_cov, rtol=2e-5, atol=1e-9)
    # and against the reference's own float32 m_step output
    np.testing.assert_allclose(mu2, g["m0_mu"], rtol=0, atol=5e-6)


@pytest.mark.parametrize("variant,cov_type", [("W", "diag"), ("W", "spherical"), ("G", "diag")])
def test_train_small(ctx, variant, cov_type):
    g = load_golden("flat_small_%s_%s.npz" % (variant, cov_type))
    X = g["X"]
    ctx.set_points(X)
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    for iters in (1, 5):
        inv, mu, w, cov, lls, conv = ctx.flat_train(iters, 0.0, g["mu0"], g["cov0"], g["w0"], cov_type, variant)
        o_inv, o_mu, o_w, o_cov, o_lls, _ = flat_em.train(f64(X), iters, 0.0, f64(g["mu0"]), f64(g["cov0"]),
                                                          f64(g["w0"]), cov_type, variant)
        assert len(lls) == iters and not conv
        np.testing.assert_allclose(lls, o_lls, rtol=0, atol=2e-5)
        np.testing.assert_allclose(mu, o_mu, rtol=0, atol=5e-6)
        np.testing.assert_allclose(w, o_w, rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(cov, o_cov, rtol=1e-4, atol=1e-9)
        np.testing.assert_allclose(inv, o_inv, rtol=1e-4)
        # the reference's float32 trajectory
        pre = "it%d_" % iters
        np.testing.assert_allclose(lls, g[pre + "lls"], rtol=0, atol=5e-5)
        np.testing.assert_allclose(mu, g[pre + "mu"], rtol=0, atol=2e-5)


@pytest.mark.parametrize("cov_type", ["diag", "spherical"])
def test_fit_far_from_the_origin(ctx, cov_type):
    """A cloud in sensor coordinates: 60 m from the origin, 40 m across, clusters of a few decimetres (J = 100, which
    takes the packed fused kernel).  The fused kernel forms its quadratic forms from x - mu (nothing is lost to the
    offset) and sums its first moments about the weighted mean of the component means (their rounding error is relative
    to the model's extent, 2^-24 x 40 m per addition).  The fitted means stay within 1e-4 m of the float64 oracle's -- 13 ulp of a
    float32 coordinate at 100 m; the parameters travel between iterations as float32 like the reference's, and that, not
    the moment sums, sets the figure: the kernel that summed first moments about each component's own mean (round 3)
    differs from the oracle by the same 4.2e-5 m in the same elements -- and the log-likelihood trace within 5e-5."""
    rs = np.random.RandomState(11)
    J, N = 100, 60000
    centres = rs.rand(J, 3) * np.array([40.0, 40.0, 4.0]) + np.array([60.0, -35.0, 1.0])
    X = (centres[rs.randint(J, size=N)] + rs.randn(N, 3) * np.array([0.3, 0.3, 0.1])).astype(np.float32)
    mu0 = X[rs.choice(N, J, replace=False)].copy()
    w0 = (np.ones(J) / J).astype(np.float32)
    cov0 = np.full((J, 3) if cov_type == "diag" else (J,), 1.0, np.float32)
    ctx.set_points(X)
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    inv, mu, w, cov, lls, conv = ctx.flat_train(8, 0.0, mu0, cov0, w0, cov_type, "W")
    o_inv, o_mu, o_w, o_cov, o_lls, _ = flat_em.train(f64(X), 8, 0.0, f64(mu0), f64(cov0), f64(w0), cov_type, "W")
    np.testing.assert_allclose(lls, o_lls, rtol=0, atol=5e-5)
    np.testing.assert_allclose(mu, o_mu, rtol=0, atol=1e-4)
    print("far from the origin (%s): max |mu - oracle| %.3g m, max |w - oracle| %.3g, max rel cov %.3g, max |lls - oracle| %.3g"
          % (cov_type, np.abs(mu - o_mu).max(), np.abs(w - o_w).max(), np.abs(cov / o_cov - 1).max(), np.abs(lls - o_lls).max()))
    np.testing.assert_allclose(w, o_w, rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(cov, o_cov, rtol=1e-3, atol=1e-8)


def test_fit_with_a_stray_component_far_outside_the_model(ctx):
    """the fused kernel's common origin for the first moments is the WEIGHTED mean of the component means.
    One component of the initial model sits 1e5 cloud extents away with a negligible weight (what a dead component at
    the coordinate origin is to a cloud in map coordinates): an unweighted mean of the means would put the origin
    40 km outside the cloud and cost every live component ~2^-24 x 40 km = 2.4 mm per addition; the fit must stay as
    close to the float64 oracle as the same fit without the stray component does (1e-4 m)."""
    rs = np.random.RandomState(12)
    J, N = 100, 60000
    centres = rs.rand(J - 1, 3) * np.array([40.0, 40.0, 4.0]) + np.array([60.0, -35.0, 1.0])
    X = (centres[rs.randint(J - 1, size=N)] + rs.randn(N, 3) * np.array([0.3, 0.3, 0.1])).astype(np.float32)
    mu0 = np.concatenate([X[rs.choice(N, J - 1, replace=False)], np.float32([[4.0e6, -4.0e6, 4.0e5]])]).astype(np.float32)
    w0 = np.concatenate([np.ones(J - 1) / (J - 1) * (1 - 1e-12), [1e-12]]).astype(np.float32)
    cov0 = np.full((J, 3), 1.0, np.float32)
    ctx.set_points(X)
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    inv, mu, w, cov, lls, conv = ctx.flat_train(6, 0.0, mu0, cov0, w0, "diag", "W")
    o_inv, o_mu, o_w, o_cov, o_lls, _ = flat_em.train(f64(X), 6, 0.0, f64(mu0), f64(cov0), f64(w0), "diag", "W")
    live = slice(0, J - 1)
    print("stray component: max |mu - oracle| over the live components %.3g m, max |lls - oracle| %.3g"
          % (np.abs(mu[live] - o_mu[live]).max(), np.abs(lls - o_lls).max()))
    np.testing.assert_allclose(lls, o_lls, rtol=0, atol=5e-5)
    np.testing.assert_allclose(mu[live], o_mu[live], rtol=0, atol=1e-4)
    np.testing.assert_allclose(w[live], o_w[live], rtol=2e-5, atol=1e-7)


def test_train_early_stop(ctx):
    g = load_golden("flat_small_W_diag.npz")
    X = g["X"]
    ctx.set_points(X)
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    tol = 0.2
    inv, mu, w, cov, lls, conv = ctx.flat_train(30, tol, g["mu0"], g["cov0"], g["w0"], "diag", "W")
    o = flat_em.train(f64(X), 30, tol, f64(g["mu0"]), f64(g["cov0"]), f64(g["w0"]), "diag", "W")
    assert conv and o[5]
    assert len(lls) == len(o[4]) < 30
    np.testing.assert_allclose(mu, o[1], rtol=0, atol=1e-5)


@pytest.mark.parametrize("variant,cov_type", FLAVOURS)
@pytest.mark.parametrize("J", [100, 800])
def test_train_bunny_20_iterations(ctx, bunny, J, variant, cov_type):
    """BASELINE config 1/2 end to end: 20 EM iterations, tol=0, same seeded init as the
    reference run behind the golden file, all three flavours."""
    g = bunny_golden(J, variant, cov_type)
    X = bunny
    mu0 = X[g["init_idx"]]
    w0 = (np.ones(J) / J).astype(np.float32)
    cov0 = (0.1 * np.ones((J, 3) if cov_type == "diag" else (J,))).astype(np.float32)
    ctx.set_points(X)
    inv, mu, w, cov, lls, conv = ctx.flat_train(20, 0.0, mu0, cov0, w0, cov_type, variant)
    assert len(lls) == 20
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    o_inv, o_mu, o_w, o_cov, o_lls, _ = flat_em.train(f64(X), 20, 0.0, f64(mu0), f64(cov0), f64(w0), cov_type, variant)
    d_ll = np.abs(lls - np.array(o_lls)).max()
    d_ref = np.abs(lls - g["lls"]).max()
    d_refref = np.abs(np.array(o_lls) - g["lls"]).max()
    print("J=%d %s/%s lls: |gpu-oracle64| %.3g  |gpu-ref32| %.3g  |ref32-oracle64| %.3g"
          % (J, variant, cov_type, d_ll, d_ref, d_refref))
    # The GPU's float32-state trajectory stays with the float64 oracle (measured 1.5e-6 .. 3e-6 for flavour W);
    # the reference's own float32 trajectory is d_refref (~1e-3) away from that oracle, and the GPU must be no
    # further from the reference than the oracle is (+ the same small margin).  Flavour G clips the covariance
    # at 0 without a floor, so its late iterations amplify float32 state rounding: bound relative to d_refref.
    tol_ll = 2e-5 if variant == "W" else max(2e-5, 0.25 * d_refref)
    assert d_ll <= tol_ll, (d_ll, tol_ll)
    assert d_ref <= d_refref + tol_ll
    live = o_w > 1e-4
    d_mu = np.abs(mu - o_mu)[live].max()
    d_mu_ref = np.abs(g["mu"] - o_mu)[live].max()
    print("J=%d means: |gpu-oracle64| %.3g  |ref32-oracle64| %.3g" % (J, d_mu, d_mu_ref))
    assert d_mu <= max(2e-5, 0.5 * d_mu_ref)
    np.testing.assert_allclose(w[live], o_w[live], rtol=0, atol=max(1e-5, 0.5 * np.abs(g["w"] - o_w).max()))


@pytest.mark.parametrize("N,J", [(1, 1), (63, 5), (65, 37), (1000, 64), (777, 257), (300, 1023), (500, 1024)])
def test_ragged_sizes(ctx, N, J):
    rs = np.random.RandomState(N * 1000 + J)
    X = rs.rand(N, 3).astype(np.float32)
    mu = rs.rand(J, 3).astype(np.float32)
    inv = (1.0 / np.sqrt(0.01 + 0.1 * rs.rand(J, 3))).astype(np.float32)
    w = rs.rand(J).astype(np.float32) + 0.01
    w /= w.sum()
    check_estep(ctx, X, inv, mu, w, "diag", "W", "ragged %dx%d" % (N, J))
    ctx.set_points(X)
    stats, sum_lpn, n = ctx.flat_stats(inv, mu, w, "diag", "W")
    o_mean, o_lr, o_lpn, _ = oracle64_estep(X, inv, mu, w, "diag", "W")
    r = np.exp(o_lr)
    assert n == N
    np.testing.assert_allclose(sum_lpn, o_lpn.sum(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(stats[:, 0], r.sum(0), rtol=1e-4, atol=1e-5)
    d = X[:, None, :].astype(np.float64) - mu[None].astype(np.float64)
    np.testing.assert_allclose(stats[:, 1:4], (r[:, :, None] * d).sum(0), rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(stats[:, 4:7], (r[:, :, None] * d * d).sum(0), rtol=1e-3, atol=2e-5)


def test_degenerate_rows(ctx):
    """Rows where every exponential underflows: the reference's normaliser is log(0 + eps);
    responsibilities are ~0 and rows do NOT sum to one (SURVEY 7 'hard parts').  Also zero
    weights under flavour G (log 0 = -inf)."""
    X = np.array([[0, 0, 0], [100, 100, 100], [0.1, 0.2, 0.3], [-50, 0, 50]], dtype=np.float32)
    mu = np.array([[0, 0, 0], [0.1, 0.2, 0.3], [1, 1, 1]], dtype=np.float32)
    inv = np.full((3, 3), 10.0, dtype=np.float32)
    w = np.array([0.5, 0.5, 0.0], dtype=np.float32)
    for variant in ("W", "G"):
        ctx.set_points(X)
        mean, lr, lpn, am = ctx.flat_estep(inv, mu, w, "diag", variant, want_lpn=True, want_argmax=True)
        with np.errstate(divide="ignore"):
            o_mean, o_lr, o_lpn, o_am = oracle64_estep(X, inv, mu, w, "diag", variant)
        np.testing.assert_allclose(lpn.get(), o_lpn, rtol=1e-5, atol=1e-5)
        assert np.abs(np.exp(lr.get().astype(np.float64)) - np.exp(o_lr)).max() < RESP_TOL
        assert np.isclose(lpn.get()[1], np.log(1e-8), atol=1e-4)
        assert np.array_equal(am.get(), o_am)


def test_full_size_properties(ctx):
    """BASELINE config 3 size (N = 1e6 uniform, J = 800): size-independent properties plus
    oracle parity on sampled rows."""
    N, J = 1_000_000, 800
    X = np.random.RandomState(0).rand(N, 3).astype(np.float32)
    idx = np.random.RandomState(100).choice(N, J, replace=False)
    mu0 = X[idx].copy()
    w0 = (np.ones(J) / J).astype(np.float32)
    cov0 = (0.1 * np.ones((J, 3))).astype(np.float32)
    ctx.set_points(X)
    # two EM iterations so the parameters are not the trivial initial ones
    inv, mu, w, cov, lls, _ = ctx.flat_train(2, 0.0, mu0, cov0, w0, "diag", "W")
    mean, lr, lpn, am = ctx.flat_estep(inv, mu, w, "diag", "W", want_lpn=True, want_argmax=True)
    lpn_h = lpn.get()
    rows = np.random.RandomState(5).choice(N, 512, replace=False)
    lr_rows = np.stack([lr.get()[r] for r in rows]) if False else lr.get()[rows]
    o_mean, o_lr, o_lpn, o_am = oracle64_estep(X[rows], inv, mu, w, "diag", "W")
    assert np.abs(np.exp(lr_rows.astype(np.float64)) - np.exp(o_lr)).max() <= RESP_TOL
    np.testing.assert_allclose(lpn_h[rows], o_lpn, rtol=1e-5, atol=1e-5)
    flips = am.get()[rows] != o_am
    if flips.any():                                   # hard assignments: identical except genuine near-ties
        part = np.partition(np.exp(o_lr)[flips], -2, axis=1)
        assert ((part[:, -1] - part[:, -2]) < RESP_TOL).all(), "label flip that is not a near-tie"
    # property 1: rows sum to 1 - eps * exp(-lpn)   (the reference's +eps normaliser)
    sums = np.exp(lr_rows.astype(np.float64)).sum(1)
    np.testing.assert_allclose(sums, 1.0 - 1e-8 * np.exp(-o_lpn), rtol=0, atol=2e-5)
    # property 2: mean of the per-point normalisers == returned scalar == lls of a 3rd iteration
    assert abs(lpn_h.astype(np.float64).mean() - mean) < 1e-5
    # property 3: the materialised path (E-step -> M-step from resp) and the fused path agree
    w_m, mu_m, cov_m = ctx.flat_mstep(lr.exp(), "diag", "W", centre_hint=mu)
    inv3, mu3, w3, cov3, lls3, _ = ctx.flat_train(3, 0.0, mu0, cov0, w0, "diag", "W")
    assert abs(lls3[2] - mean) < 2e-5
    np.testing.assert_allclose(mu_m, mu3, rtol=0, atol=2e-6)
    np.testing.assert_allclose(w_m, w3, rtol=2e-5, atol=1e-8)
    np.testing.assert_allclose(cov_m, cov3, rtol=1e-4, atol=1e-9)
    # property 4: weights sum to ~1, labels in range
    assert abs(w3.sum() - 1.0) < 1e-4
    a = am.get()
    assert a.min() >= 0 and a.max() < J


@pytest.mark.parametrize("variant,cov_type", FLAVOURS)
def test_c3_sampled_rows_at_20_iteration_parameters(ctx, variant, cov_type):
    """BASELINE config 3 (N = 1e6 uniform, J = 800) where the parameters are no longer near their initial values:
    20 EM iterations on the device, then the materialising E-step -- both of its loops -- against the float64
    oracle on 20 000 sampled rows, every flavour the reference has.  Hard assignments: identical except genuine
    near-ties (top-2 responsibilities closer than 1e-5)."""
    N, J = 1_000_000, 800
    X = np.random.RandomState(0).rand(N, 3).astype(np.float32)
    idx = np.random.RandomState(100).choice(N, J, replace=False)
    mu0 = X[idx].copy()
    w0 = (np.ones(J) / J).astype(np.float32)
    cov0 = (0.1 * np.ones((J, 3) if cov_type == "diag" else (J,))).astype(np.float32)
    ctx.set_points(X)
    inv, mu, w, cov, lls, _ = ctx.flat_train(20, 0.0, mu0, cov0, w0, cov_type, variant)
    assert len(lls) == 20 and np.isfinite(lls).all()
    assert np.abs(mu - mu0).max() > 1e-2                          # the fit has moved
    rows = np.sort(np.random.RandomState(6).choice(N, 20000, replace=False))
    o_mean, o_lr, o_lpn, o_am = oracle64_estep(X[rows], inv, mu, w, cov_type, variant)
    o_r = np.exp(o_lr)
    for want_argmax in (False, True):                             # constant-shift loop / row-maximum loop
        mean, lr, lpn, am = ctx.flat_estep(inv, mu, w, cov_type, variant, want_lpn=True, want_argmax=want_argmax)
        lr_rows = lr.get()[rows]
        lpn_h = lpn.get()
        d = np.abs(np.exp(lr_rows.astype(np.float64)) - o_r).max()
        print("C3 %s/%s argmax=%s: max|dresp| %.3g on %d rows" % (variant, cov_type, want_argmax, d, len(rows)))
        assert d <= RESP_TOL
        np.testing.assert_allclose(lpn_h[rows], o_lpn, rtol=2e-5, atol=2e-5)
        assert abs(lpn_h.astype(np.float64).mean() - mean) < 1e-5 * max(1.0, abs(mean))
        big = o_lr > -30
        np.testing.assert_allclose(lr_rows[big], o_lr[big], rtol=2e-5, atol=2e-5)
        if want_argmax:
            a = am.get()
            assert a.min() >= 0 and a.max() < J
            flips = a[rows] != o_am
            if flips.any():
                part = np.partition(o_r[flips], -2, axis=1)
                assert ((part[:, -1] - part[:, -2]) < RESP_TOL).all(), "label flip that is not a near-tie"
            print("C3 %s/%s label flips (near-ties) %d / %d" % (variant, cov_type, int(flips.sum()), len(rows)))
        del lr, lpn, am
    # the fused loop's 21st log-likelihood == the materialising E-step's mean normaliser at the 20-iteration parameters
    lls21 = ctx.flat_train(21, 0.0, mu0, cov0, w0, cov_type, variant)[4]
    assert abs(lls21[20] - mean) < 2e-5 * max(1.0, abs(mean))
    lab = ctx.flat_predict(inv, mu, w, cov_type, variant).get()
    flips = lab[rows] != o_am
    if flips.any():
        part = np.partition(o_r[flips], -2, axis=1)
        assert ((part[:, -1] - part[:, -2]) < RESP_TOL).all()


def _label_checksum(lab):
    lab = np.asarray(lab).astype(np.uint64)
    pos = np.arange(1, len(lab) + 1, dtype=np.uint64)
    return int((lab * pos).sum(dtype=np.uint64)), int((lab * lab * pos).sum(dtype=np.uint64))


@pytest.mark.parametrize("variant,cov_type", FLAVOURS)
def test_c3_training_matches_oracle_fixture(ctx, variant, cov_type):
    """BASELINE config 3 -- the configuration bench.py's headline `value` is timed on -- against the ORACLE at full
    size: tests/golden/flat_uniform1M_J800_oracle.npz holds oracle.flat_em's float64 EM (tools/gen_oracle_fixtures.py
    --only flat1m: the oracle's op sequence over 16384-row blocks, asserted equal to oracle.flat_em.train) on the
    bench frame, 3 iterations, every flavour.  Held to it: (i) the fused training loop (flat_fused_pk_kernel ->
    flat_reduce_kernel -> flat_finalize_kernel over ~500 workgroup partials) -- lls, mu, w, cov, inv_std with the
    tolerances of test_train_small; (ii) the same three iterations through the materialised e_step -> m_step loop;
    (iii) predict at the oracle's final parameters: ALL 10^6 labels equal the oracle's except on rows whose two
    largest responsibilities are closer than 1e-5 (north_star's near-tie rule) -- checked exactly through a
    position-weighted checksum over the other rows, the population per component, and 20 000 sampled labels."""
    g = load_golden("flat_uniform1M_J800_oracle.npz")
    k = "%s_%s_" % (variant, cov_type)
    N, J, iters = int(g["N"]), int(g["J"]), int(g["iters"])
    X = np.random.RandomState(int(g["cloud_seed"])).rand(N, 3).astype(np.float32)
    idx = np.random.RandomState(int(g["init_seed"])).choice(N, J, replace=False)
    assert np.array_equal(idx, g["init_idx"])
    mu0 = X[idx].copy()
    w0 = (np.ones(J) / J).astype(np.float32)
    cov0 = (0.1 * np.ones((J, 3) if cov_type == "diag" else (J,))).astype(np.float32)
    ctx.set_points(X)

    def check_fit(tag, inv, mu, w, cov, lls):
        d_ll = np.abs(np.asarray(lls, dtype=np.float64) - g[k + "lls"]).max()
        d_mu = np.abs(mu - g[k + "mu"]).max()
        print("C3 fit %s/%s %s: max|dlls| %.3g max|dmu| %.3g max rel dw %.3g max rel dcov %.3g"
              % (variant, cov_type, tag, d_ll, d_mu, np.abs(w / g[k + "w"] - 1).max(), np.abs(cov / g[k + "cov"] - 1).max()))
        assert len(lls) == iters and d_ll <= 2e-5
        assert d_mu <= 5e-6
        np.testing.assert_allclose(w, g[k + "w"], rtol=1e-5, atol=1e-10)
        np.testing.assert_allclose(cov, g[k + "cov"], rtol=1e-4, atol=1e-9)
        np.testing.assert_allclose(inv, g[k + "inv"], rtol=1e-4)
        assert abs(float(w.astype(np.float64).sum()) - float(g[k + "w"].sum())) < 1e-5

    # (i) the device-resident loop: the kernel bench.py's `value` times
    inv, mu, w, cov, lls, _ = ctx.flat_train(iters, 0.0, mu0, cov0, w0, cov_type, variant)
    check_fit("fused", inv, mu, w, cov, lls)
    # (ii) the materialised loop of a caller of the module functions: e_step -> m_step(exp(log_resp)) -> inv_cov
    inv_m = flat_em.inv_std_from_cov(cov0, variant, initial=True).astype(np.float32)
    mu_m, w_m, cov_m, lls_m = mu0, w0, cov0, []
    lr = None
    for _ in range(iters):
        mean, lr, _, _ = ctx.flat_estep(inv_m, mu_m, w_m, cov_type, variant, out=lr)
        lls_m.append(mean)
        w_m, mu_m, cov_m = ctx.flat_mstep(lr.exp(), cov_type, variant, centre_hint=mu_m)
        inv_m = flat_em.inv_std_from_cov(cov_m, variant).astype(np.float32)
    del lr
    check_fit("materialised", inv_m, mu_m, w_m, cov_m, lls_m)
    # (iii) hard labels of ALL points at the float32 roundings of the oracle's final parameters
    f32 = lambda a: np.asarray(a, dtype=np.float32)
    lab = ctx.flat_predict(f32(g[k + "inv"]), f32(g[k + "mu"]), f32(g[k + "w"]), cov_type, variant).get()
    assert lab.min() >= 0 and lab.max() < J
    near = g[k + "near_rows"]
    clear = np.ones(N, dtype=bool)
    clear[near] = False
    got = np.array(_label_checksum(np.where(clear, lab, 0)), dtype=np.uint64)
    assert np.array_equal(got, g[k + "checksum_clear"]), "a hard label differs on a row that is not near a tie"
    flips_near = int((lab[near] != g[k + "near_labels"]).sum())
    sample = g["sample"]
    flips = lab[sample] != g[k + "labels_sample"]
    assert (g[k + "gap_sample"][flips] < RESP_TOL).all()
    pop = np.bincount(lab, minlength=J)
    assert np.abs(pop - g[k + "population"]).sum() <= 2 * flips_near
    print("C3 labels %s/%s: %d of %d near-tie rows (top-2 gap < 1e-5) differ, every other of the 10^6 rows is equal"
          % (variant, cov_type, flips_near, len(near)))
    # the arg-max of the materialising E-step (row-maximum loop) sees the same values
    _, _, _, am = ctx.flat_estep(f32(g[k + "inv"]), f32(g[k + "mu"]), f32(g[k + "w"]), cov_type, variant,
                                 want_log_resp=False, want_argmax=True)
    am = am.get()
    assert np.array_equal(np.array(_label_checksum(np.where(clear, am, 0)), dtype=np.uint64), g[k + "checksum_clear"])


def test_c3_training_20_iterations_match_oracle_fixture(ctx):
    """The headline configuration once more, where the fit has left its initial parameters far behind: 20 fused EM
    iterations on the bench frame (flavour W / diag) against oracle.flat_em's float64 loop on the SAME million points
    (tests/golden/flat_uniform1M_J800_oracle_20it.npz, tools/gen_oracle_fixtures.py --only flat1m_long): the whole
    log-likelihood trace and the final model."""
    g = load_golden("flat_uniform1M_J800_oracle_20it.npz")
    N, J, iters = int(g["N"]), int(g["J"]), int(g["iters"])
    X = np.random.RandomState(int(g["cloud_seed"])).rand(N, 3).astype(np.float32)
    idx = np.random.RandomState(int(g["init_seed"])).choice(N, J, replace=False)
    assert np.array_equal(idx, g["init_idx"])
    mu0 = X[idx].copy()
    w0 = (np.ones(J) / J).astype(np.float32)
    cov0 = (0.1 * np.ones((J, 3))).astype(np.float32)
    ctx.set_points(X)
    inv, mu, w, cov, lls, _ = ctx.flat_train(iters, 0.0, mu0, cov0, w0, "diag", "W")
    d_ll = np.abs(np.asarray(lls, dtype=np.float64) - g["lls"]).max()
    live = g["w"] > 1e-5
    d_mu = np.abs(mu - g["mu"])[live].max()
    print("C3 20 iterations: max|dlls| %.3g (trace %.4f .. %.4f), max|dmu| %.3g, max rel dw %.3g, max rel dcov %.3g, live %d"
          % (d_ll, g["lls"][0], g["lls"][-1], d_mu, np.abs(w / g["w"] - 1)[live].max(), np.abs(cov / g["cov"] - 1)[live].max(),
             int(live.sum())))
    # (measured: 5e-7 / 4e-7 / 5e-6 / 9e-6 -- the bounds leave a factor of ten)
    assert len(lls) == iters and d_ll <= 5e-6
    assert np.abs(mu - mu0).max() > 1e-2                          # the fit has moved
    assert d_mu <= 5e-6
    np.testing.assert_allclose(w[live], g["w"][live], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(cov[live], g["cov"][live], rtol=2e-4, atol=1e-9)
    np.testing.assert_allclose(inv[live], g["inv"][live], rtol=1e-4)


def test_store_pacer_controller_backs_off(monkeypatch):
    """The materialising E-step offers its rows at a controlled rate (StorePacer / PaceCtl, csrc/flat_kernels.hip): a
    context started far above the write path's knee (HGMM_PACE_START=7800) must walk down -- launches that run > 6 %
    longer than the rate explains are strikes, three in a row lower it by 2 % -- without ever waiting for its evidence,
    and the table it writes is the same whatever the rate; a fixed rate (HGMM_ESTEP_TARGET_GBS) switches the controller
    off; small tables are not judged.  (Probing upwards -- 2 % on probation after 24 clean launches -- cannot happen above
    PACE_CEILING_GBS = 7600, so `ups` stays 0 here.)"""
    import hgmm_amd
    N, J = 1_000_000, 800
    X = np.random.RandomState(0).rand(N, 3).astype(np.float32)
    idx = np.random.RandomState(100).choice(N, J, replace=False)
    mu, w = X[idx].copy(), (np.ones(J) / J).astype(np.float32)
    inv = (1 / np.sqrt(0.01 * np.ones((J, 3)))).astype(np.float32)
    rows = np.arange(0, N, 20011)
    monkeypatch.setenv("HGMM_PACE_START", "7800")        # (options start from the environment: read once, at creation)
    c = hgmm_amd.Context(0)
    monkeypatch.delenv("HGMM_PACE_START")
    try:
        assert c.config_get("pace_start") == 7800
        c.set_points(X)
        lr = c.empty((N, J), np.float32)
        mean0 = c.flat_estep(inv, mu, w, out=lr)[0]
        ref = lr.get()[rows].copy()
        assert c.pace_info() == (7800.0, 0, 0)
        for _ in range(20):
            c.flat_estep(inv, mu, w, out=lr, lazy_mean=True)            # nobody waits: the evidence is only ever queried
        c.synchronize()                                                  # (... so launches enqueued far ahead mostly go unobserved)
        for _ in range(40):
            mean1 = c.flat_estep(inv, mu, w, out=lr)[0]
        target, steps, ups = c.pace_info()
        print("store pacer: 7800 -> %.0f GB/s after %d steps down in 61 launches" % (target, steps))
        assert steps >= 2 and ups == 0 and 5800.0 <= target < 7800.0 * 0.98 ** 2 + 1.0
        assert np.array_equal(lr.get()[rows], ref) and mean1 == mean0           # the rate does not touch the values
        # a small table (4 MB) is neither paced down nor judged
        c.set_points(X[:5000])
        small = c.empty((5000, J), np.float32)
        for _ in range(8):
            c.flat_estep(inv, mu, w, out=small)
        assert c.pace_info() == (target, steps, ups)
        del small, lr
    finally:
        c.close()
    c = hgmm_amd.Context(0).config_set("estep_target_gbs", 6000)
    try:
        assert c.config_get("pace_start") == 6600
        c.set_points(X)
        lr = c.empty((N, J), np.float32)
        for _ in range(12):
            c.flat_estep(inv, mu, w, out=lr)
        assert c.pace_info() == (6000.0, 0, 0)
        assert np.array_equal(lr.get()[rows], ref)
    finally:
        c.close()


def test_store_pacer_recovers_after_a_congested_phase(monkeypatch):
    """What the controller learns under congestion must not bind it for life.  Phase 1: a second context on
    the same GPU streams pure stores (util_fill) from its own thread while this one runs materialising E-steps -- their
    launches run long, the rate steps down and the failed rate is remembered as a ceiling.  Phase 2: the other stream
    stops; the ceiling is forgotten after HGMM_PACE_FORGET clean launches (40 here, 10 000 by default) and probes on
    probation bring the rate back to where it started (within one 2 % step) or beyond.  Phase 3: the same congestion
    again, then hgmm_pace_reset(): rate, ceiling and counters are back at their initial values at once.  The table is the
    same whatever the rate."""
    import threading
    import hgmm_amd
    N, J = 1_000_000, 800
    X = np.random.RandomState(0).rand(N, 3).astype(np.float32)
    idx = np.random.RandomState(100).choice(N, J, replace=False)
    mu, w = X[idx].copy(), (np.ones(J) / J).astype(np.float32)
    inv = (1 / np.sqrt(0.01 * np.ones((J, 3)))).astype(np.float32)
    rows = np.arange(0, N, 20011)
    c, other = hgmm_amd.Context(0).config_set("pace_forget", 40), hgmm_amd.Context(0)
    try:
        c.set_points(X)
        lr = c.empty((N, J), np.float32)
        hog_buf = other.empty((N, J), np.float32)
        for _ in range(6):
            c.flat_estep(inv, mu, w, out=lr)
        ref = lr.get()[rows].copy()
        start, down0, _ = c.pace_info()
        assert down0 == 0

        def congested(launches):
            stop = threading.Event()

            def hog():
                while not stop.is_set():
                    for _ in range(4):
                        other.util_fill(hog_buf, 1.0, True, 0, 1)
                    other.synchronize()
            t = threading.Thread(target=hog)
            t.start()
            try:
                for _ in range(launches):
                    c.flat_estep(inv, mu, w, out=lr)
            finally:
                stop.set()
                t.join(60)
            return c.pace_info()

        t1, d1, u1 = congested(60)
        print("store pacer under a competing store stream: %.0f -> %.0f GB/s, %d steps down" % (start, t1, d1))
        assert d1 >= 1 and t1 < start
        for _ in range(400):
            c.flat_estep(inv, mu, w, out=lr)
        t2, d2, u2 = c.pace_info()
        print("... and after 400 quiet launches: %.0f GB/s (%d probes held, %d steps down in all)" % (t2, u2, d2))
        assert u2 >= 1 and t2 >= 0.979 * start
        assert np.array_equal(lr.get()[rows], ref)
        t3, d3, _ = congested(40)
        assert t3 < t2
        c.pace_reset()
        assert c.pace_info() == (start, 0, 0)
        c.flat_estep(inv, mu, w, out=lr)
        assert c.pace_info()[0] == start and np.array_equal(lr.get()[rows], ref)
        del lr, hog_buf
    finally:
        other.close()
        c.close()


def test_profiler_reports_kernel_time(ctx):
    X = np.random.RandomState(1).rand(20000, 3).astype(np.float32)
    ctx.set_points(X)
    mu, w, cov = flat_em.seeded_init(X, 64, 3)
    ctx.profile_reset()
    ctx.profile_enable(True)
    ctx.flat_train(4, 0.0, mu, cov, w, "diag", "W")
    ctx.profile_enable(False)
    ms, n = ctx.profile_get("flat_fused")
    assert n == 4 and ms > 0


def test_rccl_single_rank_communicator(ctx, bunny):
    """The N>1 data path with world_size = 1 (the GPU box has one GPU): attaching an RCCL
    communicator routes every statistics buffer through ncclAllReduce on the context's stream;
    results must be identical to the communicator-less run."""
    import hgmm_amd
    from oracle import hgmm_tree
    X = bunny[::4]
    mu0, w0, cov0 = flat_em.seeded_init(X, 48, 2)
    ctx.set_points(X)
    ref = ctx.flat_train(6, 0.0, mu0, cov0, w0, "diag", "W")
    c2 = hgmm_amd.Context(0)
    try:
        c2.comm_init(1, 0, hgmm_amd.Context.comm_unique_id())
        c2.set_points(X)
        got = c2.flat_train(6, 0.0, mu0, cov0, w0, "diag", "W")
        for a, b in zip(ref[:5], got[:5]):
            assert np.array_equal(a, b)
        assert float(c2.allreduce([3.5], op="max")[0]) == 3.5
        P = X[:3000].astype(np.float64)
        T = hgmm_tree.n_total(2)
        idx = np.random.RandomState(1).randint(T, size=T)
        t1 = ctx.set_points(P).tree_build(2, 20.0, 1e-4, P[idx], 0.001, 100)
        t2 = c2.set_points(P).tree_build(2, 20.0, 1e-4, P[idx], 0.001, 100)
        # (tables, leaf assignment and iteration counts bit for bit; the q trace of LEVEL 0 to the last few bits only: round 6's
        #  communicator-less build takes level 0's q out of the E-step's own eight terms, a communicator keeps the separate
        #  log-likelihood kernel -- tests/test_tree_gpu.py::test_stop_rule_in_the_next_launch_equals_the_ticketed_tail)
        for a, b in zip(t1[:5], t2[:5]):
            assert np.array_equal(a, b)
        n0 = int(t1[4][0])
        np.testing.assert_allclose(t1[5], t2[5], rtol=1e-13, atol=0)
        assert np.array_equal(t1[5][n0:], t2[5][n0:])
        c2.comm_destroy()
    finally:
        c2.close()


@pytest.mark.parametrize("cov_type", ["diag", "spherical"])
def test_estimate_log_prob(ctx, cov_type):
    """estimate_log_prob[_spherical] through the drop-in module API."""
    import hgmm_amd
    from hgmm_amd.gmm_waymo import gmm_impl
    hgmm_amd.set_default_context(ctx)
    g = load_golden("flat_small_W_%s.npz" % cov_type)
    X = g["X"]
    fn = gmm_impl.estimate_log_prob if cov_type == "diag" else gmm_impl.estimate_log_prob_spherical
    lp = fn(X, g["it5_inv"], g["it5_mu"]).get()
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    ofn = flat_em.log_gauss_diag if cov_type == "diag" else flat_em.log_gauss_spherical
    o = ofn(f64(X), f64(g["it5_inv"]), f64(g["it5_mu"]))
    np.testing.assert_allclose(lp, o, rtol=2e-5, atol=2e-5)


def test_dropin_module_api(ctx, bunny):
    """The reference-shaped entry points end to end: GMM_GPU.init/compute/predict,
    train_gmm/e_step/m_step/predict with host arrays and with a resident DevicePoints."""
    import hgmm_amd
    from hgmm_amd.gmm_waymo import gmm as Wg, gmm_impl as W
    from hgmm_amd.gmmreg_gpu import gmm as Gg
    hgmm_amd.set_default_context(ctx)
    X = bunny[::4]
    np.random.seed(0)
    f = Wg.GMM_GPU(n_gmm_components=20, max_iter=5, tol=1e-9, cov_type='spherical')
    f.init()
    means, weights, covs, inv = f.compute(X)
    assert means.shape == (20, 3) and covs.shape == (20,) and inv.shape == (20,)
    labels = f.predict(X)
    assert labels.dtype == np.int64 and labels.shape == (len(X),)
    o = flat_em.predict(X.astype(np.float64), inv.astype(np.float64), means.astype(np.float64),
                        weights.astype(np.float64), 'spherical', 'W')
    assert (labels != o).mean() < 1e-3
    assert len(Wg.GMM_CPU(5, max_iter=2).__class__.__mro__) > 1
    c = Wg.GMM_CPU(n_gmm_components=5, max_iter=2)
    c.init()
    assert len(c.compute(X)) == 3
    gq = Gg.GMM_GPU(n_gmm_components=6, max_iter=3)
    gq.init()
    m2, w2 = gq.compute(X)
    assert m2.shape == (6, 3) and abs(w2.sum() - 1) < 1e-3
    # function level, resident points
    dX = W.asarray(X)
    mu0, w0, cov0 = flat_em.seeded_init(X, 16, 4)
    inv0 = 1 / np.sqrt(cov0)
    ll, lr = W.e_step(dX, inv0, mu0, w0)
    wts, mus, cvs = W.m_step(dX, lr.exp(), centre_hint=mu0)
    o_ll, o_lr = flat_em.e_step(X.astype(np.float64), inv0.astype(np.float64), mu0.astype(np.float64), w0.astype(np.float64))
    o_w, o_mu, o_cov = flat_em.m_step(X.astype(np.float64), np.exp(o_lr))
    assert abs(ll - o_ll) < 1e-5
    np.testing.assert_allclose(mus, o_mu, atol=5e-6)
    np.testing.assert_allclose(cvs, o_cov, rtol=1e-4)
    # m_step with a host responsibilities array (the reference's calling convention)
    wts2, mus2, cvs2 = W.m_step(X, np.exp(lr.get()))
    np.testing.assert_allclose(mus2, o_mu, atol=5e-6)


def test_beyond_int32_elements(ctx):
    """N x J > 2^31 elements (3.0M x 800 = 2.4e9 floats, 9.6 GB resident): 64-bit addressing in the
    E-step / M-step kernels, checked on rows from the far end of the buffer."""
    N, J = 3_000_000, 800
    rs = np.random.RandomState(9)
    X = rs.rand(N, 3).astype(np.float32)
    mu = X[rs.choice(N, J, replace=False)].copy()
    inv = (1.0 / np.sqrt(0.002 + 0.004 * rs.rand(J, 3))).astype(np.float32)
    w = (np.ones(J) / J).astype(np.float32)
    ctx.set_points(X)
    mean, lr, lpn, am = ctx.flat_estep(inv, mu, w, "diag", "W", want_lpn=True, want_argmax=True)
    assert lr.size > 2 ** 31
    for lo in (0, N // 2 + 12345, N - 300):
        rows = lr.get_rows(lo, lo + 300)
        o_mean, o_lr, o_lpn, o_am = oracle64_estep(X[lo:lo + 300], inv, mu, w, "diag", "W")
        assert np.abs(np.exp(rows.astype(np.float64)) - np.exp(o_lr)).max() <= RESP_TOL
        np.testing.assert_allclose(lpn.get_rows(lo, lo + 300), o_lpn, rtol=1e-5, atol=1e-5)
        assert (am.get_rows(lo, lo + 300) != o_am).sum() <= 1
    # M-step over the whole 9.6 GB matrix agrees with the fused statistics path
    w_m, mu_m, cov_m = ctx.flat_mstep(lr.exp(), "diag", "W", centre_hint=mu)
    stats, sum_lpn, n = ctx.flat_stats(inv, mu, w, "diag", "W")
    nk = stats[:, 0] + 1e-8
    np.testing.assert_allclose(w_m, nk / N, rtol=2e-5)
    np.testing.assert_allclose(mu_m, mu + stats[:, 1:4] / nk[:, None], rtol=0, atol=2e-6)
    lr.free()


def test_streaming_refit_harness(ctx):
    """Headless run_gmm_waymo-style loop: refit every k frames, label every frame."""
    import hgmm_amd
    from hgmm_amd.gmm_waymo.run_gmm_stream import run_stream
    hgmm_amd.set_default_context(ctx)
    rs = np.random.RandomState(0)
    centres = rs.rand(12, 3) * 10
    frames = [(centres[rs.randint(12, size=3000)] + 0.2 * rs.randn(3000, 3) + 0.01 * i) for i in range(7)]
    res = run_stream(frames, n_components=12, max_iter=15, cov_type='spherical', fit_every=3)
    assert res["frames"] == 7 and len(res["fit_s"]) == 3 and res["fps"] > 0
    assert all(l.shape == (3000,) and l.min() >= 0 and l.max() < 12 for l in res["labels"])


def test_error_paths(ctx):
    """Argument / state errors surface as HgmmError with a message (no silent fallbacks)."""
    import hgmm_amd
    c = hgmm_amd.Context(0)
    try:
        mu, w, cov = np.zeros((4, 3), np.float32), np.ones(4, np.float32) / 4, np.ones((4, 3), np.float32)
        with pytest.raises(hgmm_amd.HgmmError, match="set_points"):
            c.flat_train(2, 0.0, mu, cov, w)
        with pytest.raises(hgmm_amd.HgmmError, match="positive"):
            c.set_points(np.zeros((0, 3), np.float32))
        with pytest.raises(ValueError):
            c.set_points(np.zeros((5, 2), np.float32))
        c.set_points(np.random.RandomState(0).rand(100, 3).astype(np.float32))
        big = 16385
        with pytest.raises(hgmm_amd.HgmmError, match="outside the supported range"):
            c.flat_train(1, 0.0, np.zeros((big, 3), np.float32), np.ones((big, 3), np.float32), np.ones(big, np.float32))
        with pytest.raises(hgmm_amd.HgmmError, match="diag-only"):
            c.flat_estep(np.ones(4, np.float32), mu, w, "spherical", "G")
        with pytest.raises(ValueError):
            c.flat_estep(np.ones((4, 2), np.float32), mu, w, "diag", "W")
        with pytest.raises(hgmm_amd.HgmmError, match="no tree"):
            c.tree_reg_estep(8)
        with pytest.raises(hgmm_amd.HgmmError, match="1..6"):
            c.set_points(np.random.rand(50, 3)).tree_build(7, 1.0, 1e-4, np.zeros((8 * (8 ** 7 - 1) // 7, 3)), 0.01)
        with pytest.raises(ValueError):
            c.tree_build(2, 1.0, 1e-4, np.zeros((5, 3)), 0.01)
    finally:
        c.close()
    with pytest.raises(hgmm_amd.HgmmError, match="closed"):
        c.flat_train(2, 0.0, mu, cov, w)


@pytest.mark.parametrize("N,J", [(3000, 1025), (2500, 2000), (1500, 4096), (700, 833)])
def test_large_component_counts_chunked_path(ctx, N, J):
    """J > 1024 runs the chunked path (832-component chunks, normaliser assembled across chunks);
    J = 833 (two chunks, the second with one component) is forced through it via a 1025+ sibling
    test and exercised here through the regular path for comparison."""
    rs = np.random.RandomState(J)
    X = rs.rand(N, 3).astype(np.float32)
    mu = rs.rand(J, 3).astype(np.float32)
    inv = (1.0 / np.sqrt(0.003 + 0.02 * rs.rand(J, 3))).astype(np.float32)
    w = rs.rand(J).astype(np.float32) + 0.01
    w /= w.sum()
    lr = check_estep(ctx, X, inv, mu, w, "diag", "W", "chunked %dx%d" % (N, J))
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    # predict
    ctx.set_points(X)
    lab = ctx.flat_predict(inv, mu, w, "diag", "W").get()
    o_lab = flat_em.predict(f64(X), f64(inv), f64(mu), f64(w), "diag", "W")
    assert (lab != o_lab).sum() <= 1
    # estimate_log_prob
    lp = ctx.flat_log_prob(inv, mu, "diag").get()
    np.testing.assert_allclose(lp, flat_em.log_gauss_diag(f64(X), f64(inv), f64(mu)), rtol=2e-5, atol=2e-5)
    # M-step from the materialised matrix and fused statistics
    o_w, o_mu, o_cov = flat_em.m_step(f64(X), np.exp(f64(lr)), "diag", "W")
    w_m, mu_m, cov_m = ctx.flat_mstep(ctx.to_device(lr).exp(), "diag", "W", centre_hint=mu)
    live = o_w > 1e-6
    np.testing.assert_allclose(w_m, o_w, rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(mu_m[live], o_mu[live], rtol=0, atol=1e-5)
    np.testing.assert_allclose(cov_m[live], o_cov[live], rtol=1e-3, atol=1e-8)
    stats, sum_lpn, n = ctx.flat_stats(inv, mu, w, "diag", "W")
    assert n == N
    np.testing.assert_allclose(stats[:, 0], np.exp(f64(lr)).sum(0), rtol=1e-4, atol=1e-6)
    # training loop (3 iterations, both flavours of the stop rule are exercised elsewhere)
    cov0 = (0.05 * np.ones((J, 3))).astype(np.float32)
    w0 = (np.ones(J) / J).astype(np.float32)
    inv_t, mu_t, w_t, cov_t, lls, conv = ctx.flat_train(3, 0.0, mu, cov0, w0, "diag", "W")
    o = flat_em.train(f64(X), 3, 0.0, f64(mu), f64(cov0), f64(w0), "diag", "W")
    np.testing.assert_allclose(lls, o[4], rtol=0, atol=3e-5)
    live = o[2] > 1e-5
    np.testing.assert_allclose(mu_t[live], o[1][live], rtol=0, atol=2e-5)
    np.testing.assert_allclose(w_t, o[2], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(cov_t[live], o[3][live], rtol=2e-3, atol=1e-7)
    # early stop also works on the chunked path
    if J > 1024:
        r = ctx.flat_train(25, 0.05, mu, cov0, w0, "diag", "W")
        o2 = flat_em.train(f64(X), 25, 0.05, f64(mu), f64(cov0), f64(w0), "diag", "W")
        assert len(r[4]) == len(o2[4]) and r[5] == o2[5]


@pytest.mark.parametrize("J", [4, 64, 128, 132, 256, 300, 384, 512, 560, 640, 768, 832, 896, 1000, 1024])
def test_every_lane_layout(ctx, J):
    """One J per (vector slots, scalar slots) lane layout of the E-step / M-step kernels."""
    N = 1500 + J
    rs = np.random.RandomState(J)
    X = rs.rand(N, 3).astype(np.float32)
    mu = rs.rand(J, 3).astype(np.float32)
    inv = (1.0 / np.sqrt(0.005 + 0.05 * rs.rand(J, 3))).astype(np.float32)
    w = rs.rand(J).astype(np.float32) + 0.01
    w /= w.sum()
    lr = check_estep(ctx, X, inv, mu, w, "diag", "W", "layout J=%d" % J)
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    o_w, o_mu, o_cov = flat_em.m_step(f64(X), np.exp(f64(lr)), "diag", "W")
    ctx.set_points(X)
    w_m, mu_m, cov_m = ctx.flat_mstep(ctx.to_device(lr).exp(), "diag", "W", centre_hint=mu)
    live = o_w > 1e-6
    np.testing.assert_allclose(w_m, o_w, rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(mu_m[live], o_mu[live], rtol=0, atol=1e-5)
    np.testing.assert_allclose(cov_m[live], o_cov[live], rtol=1e-3, atol=1e-8)
    inv_t, mu_t, w_t, cov_t, lls, _ = ctx.flat_train(2, 0.0, mu, (0.05 * np.ones((J, 3))).astype(np.float32),
                                                      (np.ones(J) / J).astype(np.float32), "diag", "W")
    o = flat_em.train(f64(X), 2, 0.0, f64(mu), 0.05 * np.ones((J, 3)), np.ones(J) / J, "diag", "W")
    np.testing.assert_allclose(lls, o[4], rtol=0, atol=3e-5)
    np.testing.assert_allclose(mu_t, o[1], rtol=0, atol=2e-5)


def test_train_far_points_and_mixed_scales(ctx):
    """The training kernel shifts the log-sum-exp by a constant (the largest component constant)
    instead of the row maximum: rows far from every component (all exponentials tiny or zero) and
    a mix of very tight and very broad components must still match the float64 oracle."""
    rs = np.random.RandomState(21)
    centres = rs.rand(6, 3)
    X = (centres[rs.randint(6, size=3000)] + 0.01 * rs.randn(3000, 3)).astype(np.float32)
    X[:40] += rs.choice([-1, 1], size=(40, 3)) * rs.uniform(3, 60, size=(40, 3))     # outliers
    J = 70
    mu0 = X[rs.choice(len(X), J, replace=False)].copy()
    cov0 = (10.0 ** rs.uniform(-5.5, 0.5, size=(J, 3))).astype(np.float32)        # sigma 2e-3 .. 1.8
    w0 = rs.dirichlet(np.ones(J)).astype(np.float32)
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    ctx.set_points(X)
    for variant in ("W", "G"):
        inv, mu, w, cov, lls, _ = ctx.flat_train(3, 0.0, mu0, cov0, w0, "diag", variant)
        o = flat_em.train(f64(X), 3, 0.0, f64(mu0), f64(cov0), f64(w0), "diag", variant)
        np.testing.assert_allclose(lls, o[4], rtol=2e-6, atol=2e-5)
        np.testing.assert_allclose(w, o[2], rtol=2e-5, atol=1e-8)
        np.testing.assert_allclose(mu, o[1], rtol=0, atol=2e-5 * np.abs(X).max())
        np.testing.assert_allclose(cov, o[3], rtol=2e-4, atol=1e-9)


def test_estep_constant_shift_loop_far_points_mixed_scales_and_fallback(ctx):
    """The materialising E-step's constant-shift loop (taken when no arg-max is asked for) on rows far from every
    component and on a mix of very tight and very broad components; and a table whose largest constant is out of
    the loop's range (sigma ~ 3e-8), which must fall back to the row-maximum loop transparently."""
    rs = np.random.RandomState(23)
    centres = rs.rand(6, 3)
    X = (centres[rs.randint(6, size=2000)] + 0.01 * rs.randn(2000, 3)).astype(np.float32)
    X[:40] += rs.choice([-1, 1], size=(40, 3)) * rs.uniform(3, 60, size=(40, 3))     # outliers
    J = 70
    mu = X[rs.choice(len(X), J, replace=False)].copy()
    cov = (10.0 ** rs.uniform(-5.5, 0.5, size=(J, 3))).astype(np.float32)
    w = rs.dirichlet(np.ones(J)).astype(np.float32)
    for variant in ("W", "G"):
        inv = flat_em.inv_std_from_cov(cov, variant, initial=True)
        check_estep(ctx, X, inv, mu, w, "diag", variant, "far points %s" % variant)
    # out-of-range constants: both calls must run the same (row-maximum) loop -> bit-identical outputs
    X2 = (np.repeat(rs.rand(5, 3), 40, axis=0) + 1e-8 * rs.randn(200, 3)).astype(np.float32)
    mu2 = X2[::40].copy()
    inv2 = np.full((5, 3), 3e7, np.float32)
    w2 = np.full(5, 0.2, np.float32)
    ctx.set_points(X2)
    _, a, la, _ = ctx.flat_estep(inv2, mu2, w2, "diag", "G", want_lpn=True, want_argmax=False)
    _, b, lb, _ = ctx.flat_estep(inv2, mu2, w2, "diag", "G", want_lpn=True, want_argmax=True)
    a, b, la, lb = a.get(), b.get(), la.get(), lb.get()
    assert np.isfinite(la).all()
    assert np.array_equal(a, b) and np.array_equal(la, lb)


def test_train_huge_shift_takes_the_row_maximum_variant(ctx):
    """Component constants above 2^60 (sigma ~ 4e-8, reachable only with flavour G's clipped
    covariance) are outside the constant-shift kernel's range; the row-maximum variant must take
    over transparently."""
    rs = np.random.RandomState(22)
    centres = rs.rand(5, 3).astype(np.float32)
    X = np.repeat(centres, 40, axis=0) + (1e-8 * rs.randn(200, 3)).astype(np.float32)
    X = X.astype(np.float32)
    mu0 = centres.copy()
    cov0 = np.full((5, 3), 1e-15, dtype=np.float32)
    w0 = np.full(5, 0.2, dtype=np.float32)
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    ctx.set_points(X)
    inv, mu, w, cov, lls, _ = ctx.flat_train(1, 0.0, mu0, cov0, w0, "diag", "G")
    o = flat_em.train(f64(X), 1, 0.0, f64(mu0), f64(cov0), f64(w0), "diag", "G")
    assert np.isfinite(lls).all() and np.isfinite(mu).all()
    np.testing.assert_allclose(w, o[2], rtol=1e-5)
    np.testing.assert_allclose(mu, o[1], rtol=0, atol=1e-6)
    # the oracle's expanded quadratic form (the reference's) cancels ~2.5e14-sized terms here:
    # even in float64 its q carries ~0.03 of rounding noise, so only a loose check on lls
    np.testing.assert_allclose(lls, o[4], rtol=0, atol=0.05)


def test_several_resident_clouds_per_context(ctx, bunny):
    """Round 4 (VERDICT r3 missing 3): any number of clouds stay resident on a context, like any number of
    ``cupy.asarray(frame)`` arrays under the reference (hgmm_points_create / bind / destroy).  A DevicePoints taken
    earlier stays valid through every other upload -- other DevicePoints, ctx.set_points, the HGMM entry points'
    own uploads -- and using it binds it (a pointer swap).  (Rounds 1-3: a second upload invalidated the first.)"""
    import hgmm_amd
    from hgmm_amd.gmm_waymo import gmm_impl as W
    from hgmm_amd.hgmm import hgmm_gpu as H
    hgmm_amd.set_default_context(ctx)
    clouds = [bunny[k::8] + np.float32(0.01 * k) for k in range(4)]          # four different frames, different sizes
    clouds[2] = clouds[2][:-37]
    devs = [W.asarray(X) for X in clouds]
    mu0, w0, cov0 = flat_em.seeded_init(clouds[0], 8, 2)
    inv0 = (1 / np.sqrt(cov0)).astype(np.float32)
    want = [flat_em.predict(X.astype(np.float64), inv0.astype(np.float64), mu0.astype(np.float64), w0.astype(np.float64))
            for X in clouds]
    fits = [W.train_gmm(X, 3, 0.0, mu0, cov0, w0) for X in clouds]           # host arrays: the context's own cloud each time
    other = bunny[1::16].astype(np.float64)
    H.buildGMMTree(other, 1, 80.0, 1e-4, sig2=0.00034)                        # ... and the HGMM path's own upload
    for order in ((3, 0, 2, 1), (1, 1, 3, 0)):                                # any order, repeated use
        for k in order:
            lab = W.predict(devs[k], inv0, mu0, w0)
            assert isinstance(lab, hgmm_amd.DeviceArray) and lab.dtype == np.int32 and lab.shape == (len(clouds[k]),)
            assert (np.asarray(lab) != want[k]).sum() <= 2                    # (near-ties only)
            f = W.train_gmm(devs[k], 3, 0.0, mu0, cov0, w0)
            for a, b in zip(f[:4], fits[k][:4]):
                assert np.array_equal(a, b)                                   # resident == freshly uploaded, bitwise
            ll, lr = W.e_step(devs[k], inv0, mu0, w0)
            assert lr.shape == (len(clouds[k]), 8)
    # host-array predict keeps the reference's NumPy semantics: int64 on the host
    lab_h = W.predict(clouds[1], inv0, mu0, w0)
    assert isinstance(lab_h, np.ndarray) and lab_h.dtype == np.int64 and np.array_equal(lab_h, np.asarray(W.predict(devs[1], inv0, mu0, w0)))
    ctx.set_points(clouds[0])                                                 # direct upload on the context: the handles stay
    assert ctx.num_points == len(clouds[0])
    assert np.array_equal(np.asarray(W.predict(devs[2], inv0, mu0, w0)), np.asarray(W.predict(clouds[2], inv0, mu0, w0)))
    # freeing the bound cloud leaves nothing bound; the others live on; a freed handle says so
    devs[2].bind()
    devs[2].free()
    with pytest.raises(hgmm_amd.HgmmError):
        ctx.flat_predict(inv0, mu0, w0)
    with pytest.raises(RuntimeError):
        W.predict(devs[2], inv0, mu0, w0)
    assert (np.asarray(W.predict(devs[0], inv0, mu0, w0)) != want[0]).sum() <= 2
    # the class API on resident clouds: without an explicit init the reference's initialiser (which samples from the HOST
    # array, gmm_impl.py:26-41) is fed one download of the resident rows -- the same draw as on the host array
    from hgmm_amd.gmm_waymo import gmm as Wg
    clf = Wg.GMM_GPU_Base(8, max_iter=3, tol=0.0)
    clf._verbose = False
    assert np.array_equal(devs[0].get(), clouds[0].astype(np.float32))
    np.random.seed(11)
    clf.fit(devs[0])
    drawn = clf.means_.copy()
    np.random.seed(11)
    clf.fit(clouds[0])
    assert np.array_equal(clf.means_, drawn)
    clf.fit(devs[0], init=(mu0, w0, cov0))
    assert np.array_equal(clf.means_, fits[0][1])
    assert isinstance(clf.predict(devs[1]), hgmm_amd.DeviceArray) and clf.predict(clouds[1]).dtype == np.int64
    for d in devs:
        d.free()


def test_streaming_harness_matches_the_oracle_on_the_reference_frames(ctx):
    """the loop of run_gmm_waymo_gpu.py:32-61 (refit every k frames, predict every frame) on the five
    Waymo frames the reference ships.  Replayed with the oracle: same RNG stream -> same initial parameters ->
    float64 EM; the refit models must agree with it, and EVERY frame's labels must be the oracle's predict under
    the model in force (flips only at genuine near-ties)."""
    import hgmm_amd
    from hgmm_amd.gmm_waymo.run_gmm_stream import run_stream
    from hgmm_amd.gmm_waymo.gmm_impl import init_gmm_params
    hgmm_amd.set_default_context(ctx)
    fr = load_golden("waymo_frames.npz")
    frames = [fr["waymo%d" % k] for k in (1, 2, 5, 10, 50)]
    K, iters, every = 50, 50, 2                                  # the driver's NUM_COMPONENTS / MAX_ITER; refits 0, 2, 4
    res = run_stream(frames, n_components=K, max_iter=iters, cov_type='spherical', fit_every=every, tol=1e-4, seed=5)
    assert res["frames"] == 5 and len(res["models"]) == 3
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    np.random.seed(5)                                            # run_gmm_waymo_gpu.py:30
    for i, pts in enumerate(frames):
        X = pts.astype(np.float32)
        means, weights, covs, inv = res["models"][i // every]
        if i % every == 0:
            mu0, w0, cov0 = init_gmm_params(X, K, cov_type='spherical')        # same draw as inside the harness
            o = flat_em.train(f64(X), iters, 1e-4, f64(mu0), f64(cov0), f64(w0), "spherical", "W")
            o_inv, o_mu, o_w, o_cov, o_lls, o_conv = o
            lls = np.asarray(res_lls(hgmm_amd, X, mu0, cov0, w0, iters))
            # same number of iterations up to the tol = 1e-4 stop (a change of 1e-4 can fall either side by one)
            assert abs(len(lls) - len(o_lls)) <= 1
            m = min(len(lls), len(o_lls))
            np.testing.assert_allclose(lls[:m], np.array(o_lls)[:m], rtol=0, atol=5e-5)
            if len(lls) == len(o_lls):
                live = o_w > 1e-4
                np.testing.assert_allclose(means[live], o_mu[live], rtol=0, atol=2e-3 * np.abs(X).max())
                np.testing.assert_allclose(weights[live], o_w[live], rtol=0, atol=2e-4)
        lab = res["labels"][i]
        assert lab.shape == (len(X),) and lab.dtype == np.int64
        ref = flat_em.predict(f64(X), f64(inv), f64(means), f64(weights), "spherical", "W")
        bad = np.flatnonzero(lab != ref)
        if len(bad):
            _, lr, _, _ = flat_em.e_step_full(f64(X)[bad], f64(inv), f64(means), f64(weights), "spherical", "W")
            assert flat_em.near_tie_mask(lr, 1e-5).all(), (i, len(bad))
        assert len(bad) <= 1e-3 * len(X)


def res_lls(hgmm_amd, X, mu0, cov0, w0, iters):
    """log-likelihood trace of the engine's fit from explicit initial parameters (what GMM_GPU_Base.fit stores)."""
    from hgmm_amd.gmm_waymo.gmm import GMM_GPU_Base
    b = GMM_GPU_Base(len(mu0), max_iter=iters, tol=1e-4, cov_type='spherical')
    b._verbose = False
    return b.fit(X, init=(mu0, w0, cov0)).lls


def test_async_estep_and_device_scalar(ctx, bunny):
    """hgmm_flat_estep_async: nothing waits for the kernel, the host arrays may be reused at once (they were copied to
    the pinned ring), the mean log-normaliser is a DeviceScalar read on demand; results equal the blocking call's.
    Then the API-faithful loop (e_step -> m_step -> inv_cov on the host) against the oracle's."""
    import hgmm_amd
    X = bunny[::3]
    J = 100
    mu0, w0, cov0 = flat_em.seeded_init(X, J, 5)
    inv0 = (1.0 / np.sqrt(cov0)).astype(np.float32)
    ctx.set_points(X)
    mean_b, lr_b, _, _ = ctx.flat_estep(inv0, mu0, w0, "diag", "W")
    a_inv, a_mu, a_w = inv0.copy(), mu0.copy(), w0.copy()
    # (same outputs requested as from the blocking call: asking for the arg-max selects the kernel's row-maximum loop,
    #  whose log_resp differs from the constant-shift loop's in the last float32 bit)
    mean_a, lr_a, lpn_a, _ = ctx.flat_estep(a_inv, a_mu, a_w, "diag", "W", want_lpn=True, lazy_mean=True)
    a_inv[:] = np.nan; a_mu[:] = np.nan; a_w[:] = np.nan          # the call has returned: its inputs are ours again
    assert isinstance(mean_a, hgmm_amd.DeviceScalar)
    # (the blocking call adds the workgroups' partial sums on the host, the asynchronous one in a device kernel)
    assert abs(float(mean_a) - mean_b) < 1e-7 and abs(mean_a - mean_b) < 1e-7 and np.float32(mean_a) == np.float32(mean_b)
    assert np.array_equal(lr_a.get(), lr_b.get())
    assert abs(lpn_a.get().astype(np.float64).mean() - float(mean_a)) < 1e-5
    # operators: the value takes part as a NumPy float32 scalar (what xp.mean of a float32 array is in the reference);
    # non-numbers get Python's protocol instead of a TypeError, arrays take the array path
    v32 = np.float32(float(mean_a))
    assert (mean_a == None) is False and (mean_a != None) is True and (mean_a in [None, 1.0]) is False   # noqa: E711
    assert mean_a == mean_a and mean_a in [None, mean_a] and not (mean_a < mean_a)
    assert isinstance(mean_a - 1.0, np.float32) and mean_a - 1.0 == v32 - np.float32(1.0) and 2.0 * mean_a == v32 * np.float32(2.0)
    assert abs(mean_a - (-np.inf)) == np.inf                                  # `change = lower_bound - prev`, gmm_impl.py:137-139
    assert np.array_equal(mean_a * np.ones(3, np.float32), v32 * np.ones(3, np.float32))
    assert np.array_equal(np.ones(3, np.float32) * mean_a, v32 * np.ones(3, np.float32))
    dev3 = ctx.to_device(np.float32([1.0, 2.0, 3.0]))
    for prod in (mean_a * dev3, dev3 * mean_a, dev3 + mean_a, mean_a - dev3):
        assert isinstance(prod, hgmm_amd.DeviceArray)
    assert np.array_equal(np.asarray(dev3 * mean_a), np.float32([1.0, 2.0, 3.0]) * v32)
    assert np.array_equal(np.asarray(mean_a - dev3), v32 - np.float32([1.0, 2.0, 3.0]))
    with pytest.raises(TypeError):
        mean_a + "x"
    # many un-synchronised calls in a row wrap the staging ring (it synchronises before reusing a region)
    last = None
    for k in range(400):
        last = ctx.flat_estep(inv0, mu0 + np.float32(1e-4 * (k % 7)), w0, "diag", "W", want_log_resp=False, lazy_mean=True)[0]
    o_mean, _ = flat_em.e_step(X.astype(np.float64), inv0.astype(np.float64), (mu0 + np.float32(1e-4 * (399 % 7))).astype(np.float64),
                               w0.astype(np.float64), "diag", "W")
    assert abs(float(last) - o_mean) < 1e-5
    # the module-level loop of a caller of the reference's functions, 6 iterations, against the oracle's same loop
    from hgmm_amd.gmm_waymo import gmm_impl as W
    dX = W.asarray(X, ctx)
    inv, mu, w = inv0, mu0, w0
    oi, om, ow = inv0.astype(np.float64), mu0.astype(np.float64), w0.astype(np.float64)
    X64 = X.astype(np.float64)
    for _ in range(6):
        ll, lr = W.e_step(dX, inv, mu, w)
        w, mu, cov = W.m_step(dX, lr.exp(), centre_hint=mu)
        inv = (1.0 / (np.sqrt(cov + np.float32(1e-6)) + np.float32(1e-8))).astype(np.float32)
        o_ll, o_lr = flat_em.e_step(X64, oi, om, ow, "diag", "W")
        ow, om, oc = flat_em.m_step(X64, np.exp(o_lr), "diag", "W")
        oi = 1.0 / (np.sqrt(oc + 1e-6) + 1e-8)
        assert abs(ll - o_ll) < 2e-5
    np.testing.assert_allclose(mu, om, atol=2e-5)
    np.testing.assert_allclose(w, ow, rtol=1e-3, atol=1e-6)


def test_device_array_parameters_and_elementwise(ctx, bunny):
    """The reference's functions are array-module polymorphic (`xp = cupy.get_array_module(X)`, gmm_impl.py:91): CuPy
    arrays in, CuPy arrays out, nothing synchronises.  Here: responsibilities in HBM -> m_step returns DeviceArrays,
    arithmetic on them runs on the device (bitwise NumPy's float32 results for + - * / sqrt), e_step packs its table
    from them; the loop equals the host-array loop BITWISE.  DeviceScalars are pinned scalars behind an event."""
    import hgmm_amd
    from hgmm_amd.gmm_waymo import gmm_impl as W
    DA = hgmm_amd.DeviceArray
    rs = np.random.RandomState(4)
    a = (rs.rand(800, 3).astype(np.float32) + np.float32(0.01))
    b = (rs.rand(800, 3).astype(np.float32) + np.float32(0.5))
    da, db = ctx.to_device(a), ctx.to_device(b)
    e6, e8 = np.float32(1e-6), np.float32(1e-8)
    cases = [
        (da + e6, a + e6), (e6 + da, e6 + a), (da - 0.25, a - np.float32(0.25)), (1.0 - da, np.float32(1.0) - a),
        (da * 3.0, a * np.float32(3.0)), (da / 7.0, a / np.float32(7.0)), (1.0 / da, np.float32(1.0) / a),
        (np.sqrt(da), np.sqrt(a)), (da + db, a + b), (da - db, a - b), (da * db, a * b), (da / db, a / b),
        (np.maximum(da, db), np.maximum(a, b)), (np.minimum(da, 0.5), np.minimum(a, np.float32(0.5))),
        (da + b, a + b), (b / da, b / a), (-da, -a),
        (1.0 / (np.sqrt(da + e6) + e8), 1.0 / (np.sqrt(a + e6) + e8)),          # gmm_impl.py:134
    ]
    for got, want in cases:
        assert isinstance(got, DA) and got.dtype == np.float32 and got.shape == want.shape
        assert np.array_equal(np.asarray(got), want)
    np.testing.assert_allclose(np.asarray(np.exp(-da)), np.exp(-a), rtol=2e-6)
    np.testing.assert_allclose(np.asarray(np.log(da)), np.log(a), rtol=2e-6, atol=2e-7)
    # what is not a device case is answered from a host copy, with NumPy's semantics
    row = np.float32([1, 2, 3])
    assert isinstance(da * row, np.ndarray) and np.array_equal(da * row, a * row)
    assert np.array_equal(da > 0.5, a > 0.5) and np.array_equal(da ** 2, a ** 2)
    assert da.astype(np.float32, copy=False) is da and da.astype(np.float64).dtype == np.float64
    dup = da.astype(np.float32)                                   # copy=True (NumPy's / CuPy's default): a fresh device array
    assert isinstance(dup, DA) and dup is not da and dup.ptr.value != da.ptr.value and np.array_equal(np.asarray(dup), a)
    with pytest.raises(TypeError):
        da.fill(0)                                                # would act on a throw-away host copy
    assert da.sum() == a.sum() and np.array_equal(da[3], a[3]) and np.array_equal(da.T, a.T) and len(da) == 800
    # more small arrays alive at once than the arena has slabs, then released and taken again
    many = [da + float(k) for k in range(100)]
    assert all(np.array_equal(np.asarray(m), a + np.float32(k)) for k, m in enumerate(many))
    del many
    again = [da * float(k) for k in range(100)]
    assert np.array_equal(np.asarray(again[99]), a * np.float32(99))
    del again

    # ---- the loop, parameters resident vs parameters on the host ----
    X = bunny[::3]
    J = 100
    mu0, w0, cov0 = flat_em.seeded_init(X, J, 5)
    inv0 = (1.0 / np.sqrt(cov0)).astype(np.float32)
    dX = W.asarray(X, ctx)
    ll0_h, lr_h = W.e_step(dX, inv0, mu0, w0)
    rows_h = lr_h.get()
    ll0_d, lr_d = W.e_step(dX, ctx.to_device(inv0), ctx.to_device(mu0), w0)       # mixed: w is uploaded
    assert np.array_equal(lr_d.get(), rows_h) and float(ll0_d) == float(ll0_h)
    del lr_h, lr_d
    host = (inv0, mu0, w0)
    dev = (ctx.to_device(inv0), ctx.to_device(mu0), ctx.to_device(w0))
    scalars = []
    for it in range(5):
        ll_h, lr = W.e_step(dX, *host)
        w_h, mu_h, cov_h = W.m_step(dX, np.exp(lr.get()), centre_hint=host[1])      # host responsibilities -> NumPy out
        assert isinstance(mu_h, np.ndarray)
        del lr
        ll_d, lr = W.e_step(dX, *dev)
        w_d, mu_d, cov_d = W.m_step(dX, np.exp(lr), centre_hint=dev[1])             # np.exp of a big DeviceArray: lazy view
        assert isinstance(w_d, DA) and isinstance(mu_d, DA) and isinstance(cov_d, DA)
        del lr
        inv_d = 1 / (np.sqrt(cov_d + 1e-6) + 1e-8)
        assert isinstance(inv_d, DA)
        host = ((1 / (np.sqrt(cov_h + e6) + e8)).astype(np.float32), mu_h, w_h)
        dev = (inv_d, mu_d, w_d)
        scalars.append((ll_d, float(ll_h)))
        # same kernels: fused exp of the log-responsibilities vs exp on the host differ in rounding -> tolerance
        np.testing.assert_allclose(np.asarray(mu_d), mu_h, atol=2e-6)
        np.testing.assert_allclose(np.asarray(cov_d), cov_h, rtol=2e-4, atol=1e-9)
    for ll_d, ll_h in scalars:                                   # read long after they were produced
        assert abs(float(ll_d) - ll_h) < 1e-5
    # bitwise: the same loop with log_resp.exp() on both sides, device arrays vs host arrays
    def loop(device_arrays):
        p = (ctx.to_device(inv0), ctx.to_device(mu0), ctx.to_device(w0)) if device_arrays else (inv0, mu0, w0)
        lls = []
        for it in range(5):
            ll, lr = W.e_step(dX, *p)
            w, mu, cov = ctx.flat_mstep(lr.exp(), "diag", "W", centre_hint=p[1], device_out=device_arrays)
            del lr
            inv = 1 / (np.sqrt(cov + e6) + e8)
            p = (inv if device_arrays else inv.astype(np.float32), mu, w)
            lls.append(ll)
        return [np.asarray(x) for x in p], [float(v) for v in lls]
    p_d, l_d = loop(True)
    p_h, l_h = loop(False)
    assert l_d == l_h
    for x, y in zip(p_d, p_h):
        assert np.array_equal(x, y)
    # 70 scalars outstanding at once: a slot that is taken again has its previous scalar read first
    outs = [ctx.flat_estep(inv0, mu0 + np.float32(1e-4 * k), w0, "diag", "W", want_log_resp=False, lazy_mean=True)[0]
            for k in range(70)]
    vals = [float(s) for s in outs]
    for k in (0, 5, 63, 64, 69):
        ref = ctx.flat_estep(inv0, mu0 + np.float32(1e-4 * k), w0, "diag", "W", want_log_resp=False)[0]
        assert abs(vals[k] - ref) < 1e-7
    with pytest.raises(ValueError):
        ctx.flat_estep(ctx.to_device(inv0[:5]), ctx.to_device(mu0), w0, "diag", "W", want_log_resp=False)
    # predict with the parameters where they are == predict with host copies of them
    lab_d = W.predict(dX, *[ctx.to_device(x) for x in p_h])
    lab_h = W.predict(dX, *p_h)
    # (a resident cloud: the labels stay in HBM -- DeviceArray int32; NumPy int64 for host input, like the reference under NumPy)
    assert isinstance(lab_d, DA) and lab_d.dtype == np.int32 and np.array_equal(np.asarray(lab_d), np.asarray(lab_h))
    assert W.predict(X, *p_h).dtype == np.int64 and np.array_equal(W.predict(X, *p_h), np.asarray(lab_h))


def test_predict_four_row_kernel_and_array_reuse(ctx, monkeypatch):
    """predict() runs on its own arg-max kernel (four rows in flight per wave, packed arithmetic, no exponentials);
    the single-row kernel it replaced is kept for layouts it is not instantiated for: both give the same labels on
    every layout / flavour, and the labels are the oracle's up to near-ties.  Also: a DeviceArray that dies hands its
    memory to the next one of the same size (no hipMalloc / hipFree / synchronisation per call in a caller's loop)."""
    rs = np.random.RandomState(5)
    for J, ct, var, n in ((100, "diag", "G", 50001), (257, "spherical", "W", 50001), (64, "diag", "W", 3), (1000, "diag", "W", 20011),
                          (513, "diag", "G", 7777), (7, "spherical", "W", 50001), (800, "diag", "W", 30000), (1024, "diag", "W", 4097)):
        Xs = rs.rand(n, 3).astype(np.float32)
        ctx.set_points(Xs)
        mu = rs.rand(J, 3).astype(np.float32)
        w = rs.rand(J).astype(np.float32)
        w /= w.sum()
        inv = (1.0 / (0.02 + 0.1 * rs.rand(*((J, 3) if ct == "diag" else (J,))))).astype(np.float32)
        with ctx.config(predict_single_row=1):
            a = ctx.flat_predict(inv, mu, w, ct, var).get()
        b = ctx.flat_predict(inv, mu, w, ct, var).get()
        assert np.array_equal(a, b), (J, ct, var)
        o = flat_em.predict(Xs.astype(np.float64), inv.astype(np.float64), mu.astype(np.float64), w.astype(np.float64), ct, var)
        diff = np.nonzero(b != o)[0]
        if len(diff):                                  # float32 vs float64 near-ties only
            lp = (flat_em._log_gauss(Xs[diff].astype(np.float64), inv.astype(np.float64), mu.astype(np.float64), ct)
                  + flat_em._log_weights(w.astype(np.float64), var))
            r = np.arange(len(diff))
            assert np.all(np.abs(lp[r, b[diff]] - lp[r, o[diff]]) < 1e-4), (J, ct, var)
    # zero weights everywhere: index 0, like the arg-max of a constant row
    ctx.set_points(rs.rand(1000, 3).astype(np.float32))
    z = ctx.flat_predict(np.ones((8, 3), np.float32), rs.rand(8, 3).astype(np.float32), np.zeros(8, np.float32), "diag", "G").get()
    assert (z == 0).all()
    # array reuse: same size -> same memory, another size -> other memory
    a1 = ctx.empty((1 << 20,), np.float32)
    p1 = a1.ptr.value
    del a1
    a2 = ctx.empty((1 << 20,), np.float32)
    a3 = ctx.empty((1 << 20,), np.float32)
    assert a2.ptr.value == p1 and a3.ptr.value != p1
    a4 = ctx.empty(((1 << 20) + 1,), np.float32)
    assert a4.ptr.value not in (a2.ptr.value, a3.ptr.value)
