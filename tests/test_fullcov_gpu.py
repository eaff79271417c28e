"""GPU parity tests of the flat FULL-covariance EM (fp64, MFMA sufficient statistics) against the
reference's CPU twin run with n_node = J (golden) and the oracle restatement."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import hgmm_tree

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import hgmm_amd
    c = hgmm_amd.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("J", [8, 32])
def test_fullcov_matches_reference_golden(ctx, J):
    g = load_golden("fullcov_flat.npz")
    P = g["points"]
    tag = "J%d_" % J
    ctx.set_points(P)
    pi, mu, cov, labels, q = ctx.fullcov_fit(J, 80.0, 1e-4, P[g[tag + "init_idx"]], 0.00034)
    np.testing.assert_allclose(q, g[tag + "q_trace"], rtol=1e-9, atol=1e-6)
    assert np.array_equal(labels, g[tag + "current_idx"])
    np.testing.assert_allclose(pi, g[tag + "pi"], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(mu, g[tag + "mu"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(cov, g[tag + "cov"], rtol=1e-7, atol=1e-14)


@pytest.mark.parametrize("N,J", [(5032, 100), (777, 17), (64, 16), (1, 3), (3000, 800)])
def test_fullcov_vs_oracle(ctx, bunny, N, J):
    P = bunny[:: max(1, len(bunny) // N)][:N].astype(np.float64)
    rs = np.random.RandomState(J)
    idx = rs.choice(len(P), J, replace=len(P) < J)
    iters = 6 if J >= 800 else 40
    ctx.set_points(P)
    pi, mu, cov, labels, q = ctx.fullcov_fit(J, 1.0, 1e-4, P[idx], 0.0005, iters)
    o_pi, o_mu, o_cov, o_q, o_cur = hgmm_tree.build_flat_fullcov(P, J, 1.0, 1e-4, idx, 0.0005, max_iters=iters)
    assert len(q) == len(o_q)
    np.testing.assert_allclose(q, o_q, rtol=1e-9, atol=1e-6)
    assert np.array_equal(labels, o_cur)
    np.testing.assert_allclose(pi, o_pi, rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(mu, o_mu, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(cov, o_cov, rtol=1e-6, atol=1e-14)


@pytest.mark.parametrize("J", [513, 528, 540, 560, 576, 600, 737, 860, 990, 1024])
def test_fullcov_component_layouts_vs_oracle(ctx, bunny, J):
    """Every way the one-pass kernel deals the components beyond 512 out over its waves: a tail block of <= 32
    components shared by 8 / 4 / 2 / 1 waves (1, 2, 4, 8 points per half-wave), a partly filled last wave, none."""
    P = bunny[::30][:1300].astype(np.float64)
    idx = np.random.RandomState(J).choice(len(P), J, replace=False)
    ctx.set_points(P)
    pi, mu, cov, labels, q = ctx.fullcov_fit(J, 1e-30, 1e-4, P[idx], 0.0005, 3)
    o_pi, o_mu, o_cov, o_q, o_cur = hgmm_tree.build_flat_fullcov(P, J, 1e-30, 1e-4, idx, 0.0005, max_iters=3)
    np.testing.assert_allclose(q, o_q, rtol=1e-9, atol=1e-6)
    assert np.array_equal(labels, o_cur)
    np.testing.assert_allclose(pi, o_pi, rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(mu, o_mu, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(cov, o_cov, rtol=1e-6, atol=1e-14)


def test_fullcov_estep_moments_layout(ctx, bunny):
    """hgmm_fullcov_estep: the 10-float statistics expanded to the reference's m0/m1/m2 layout."""
    P = bunny[::16].astype(np.float64)
    J = 24
    rs = np.random.RandomState(1)
    idx = rs.choice(len(P), J, replace=False)
    pi = np.full(J, 1.0 / J)
    mu = P[idx].copy()
    cov = np.tile(np.identity(3) * 0.0004, (J, 1, 1))
    cov[:, 0, 1] = cov[:, 1, 0] = 0.0001            # asymmetric-looking features catch transposes
    ctx.set_points(P)
    m0, m1, m2, labels, q = ctx.fullcov_estep(pi, mu, cov)
    ok, inv, coef = hgmm_tree.node_prep(cov)
    g = pi[None, :] * hgmm_tree.pdf_pairs(P[:, None, :], mu[None], inv[None], coef[None])
    den = g.sum(1)
    gam = np.where((den > 1e-15)[:, None], g / np.where(den > 1e-15, den, 1.0)[:, None], 0.0)
    use = np.where(gam < 1e-15, 0.0, gam)
    np.testing.assert_allclose(m0, use.sum(0), rtol=1e-11)
    np.testing.assert_allclose(m1, use.T @ P, rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(m2, np.einsum('nj,na,nb->jab', use, P, P), rtol=1e-10, atol=1e-15)
    assert np.array_equal(labels, np.argmax(gam, axis=1))
    np.testing.assert_allclose(q, np.log(np.maximum(g.sum(1), 1e-15)).sum(), rtol=1e-11)


def test_fullcov_dropin_function(ctx, bunny):
    from hgmm_amd.hgmm.hgmm_gpu import fitFullCovGMM
    P = bunny[::8].astype(np.float64)
    pi, mu, cov, tr = fitFullCovGMM(P, 50, ls=5.0, sig2=0.001, ctx=ctx, return_trace=True, max_iters=30)
    assert pi.shape == (50,) and mu.shape == (50, 3) and cov.shape == (50, 3, 3)
    assert abs(pi.sum() - 1.0) < 1e-6 or (pi == 0).any()
    assert tr["labels"].min() >= 0 and tr["labels"].max() < 50


def _label_checksum(cur):
    cur = np.asarray(cur).astype(np.uint64)
    pos = np.arange(1, len(cur) + 1, dtype=np.uint64)
    return int((cur * pos).sum(dtype=np.uint64)), int((cur * cur * pos).sum(dtype=np.uint64))


def test_fullcov_1M_matches_oracle_fixture(ctx):
    """The size bench.py's `fullcov` leg times (uniform cloud N = 1e6, J = 800) against the oracle's op sequence
    run on the SAME million points (tools/gen_oracle_fixtures.py --only fullcov1m; 3 iterations): q trace,
    parameters, all 10^6 hard assignments (population per component, checksum, 20 000 sampled labels).  Then the
    size-independent properties: sum_j m0 = N, the statistics of the whole cloud == the sum of the statistics
    of 16 shards (every shard size launches other grids / tile maps), one shard against the oracle, bitwise rerun."""
    g = load_golden("fullcov_uniform1M_J800_oracle.npz")
    N, J, iters = int(g["n_points"]), int(g["J"]), int(g["max_iters"])
    P = np.random.RandomState(int(g["cloud_seed"])).rand(N, 3).astype(np.float32).astype(np.float64)
    idx = np.random.RandomState(int(g["init_seed"])).choice(N, J, replace=False)
    ctx.set_points(P)
    pi, mu, cov, labels, q = ctx.fullcov_fit(J, float(g["ls"]), float(g["ld"]), P[idx], float(g["sig2"]), iters)
    assert len(q) == iters
    np.testing.assert_allclose(q, g["q_trace"], rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(pi, g["pi"], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(mu, g["mu"], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(cov, g["cov"], rtol=1e-6, atol=1e-14)
    assert np.array_equal(labels[g["sample"]], g["labels_sample"])
    assert np.array_equal(np.bincount(labels, minlength=J), g["population"])
    assert _label_checksum(labels) == tuple(int(v) for v in g["checksum"])
    again = ctx.fullcov_fit(J, float(g["ls"]), float(g["ld"]), P[idx], float(g["sig2"]), iters)
    assert np.array_equal(again[4], q) and np.array_equal(again[3], labels) and np.array_equal(again[2], cov)
    # E-step statistics at the fitted parameters
    m0, m1, m2, lab_e, q_e = ctx.fullcov_estep(pi, mu, cov)
    assert abs(m0.sum() - N) <= 1e-9 * N
    s0, s1, s2, sq = np.zeros(J), np.zeros((J, 3)), np.zeros((J, 3, 3)), 0.0
    bounds = np.linspace(0, N, 17).astype(int)
    bounds[1] += 37                                              # ragged shard sizes
    bounds[5] -= 1001
    for a, b in zip(bounds[:-1], bounds[1:]):
        ctx.set_points(P[a:b])
        t0, t1, t2, lab_s, q_s = ctx.fullcov_estep(pi, mu, cov)
        assert np.array_equal(lab_s, lab_e[a:b])
        s0 += t0; s1 += t1; s2 += t2; sq += q_s
    np.testing.assert_allclose(s0, m0, rtol=1e-11)
    np.testing.assert_allclose(s1, m1, rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(s2, m2, rtol=1e-10, atol=1e-9)
    assert abs(sq - q_e) <= 1e-11 * abs(q_e)
    # the first 20 000 points as their own cloud against the oracle's E-step
    Ps = P[:20000]
    ctx.set_points(Ps)
    t0, t1, t2, lab_s, q_s = ctx.fullcov_estep(pi, mu, cov)
    ok, inv, coef = hgmm_tree.node_prep(cov)
    gm = pi[None, :] * hgmm_tree.pdf_pairs(Ps[:, None, :], mu[None], inv[None], coef[None])
    den = gm.sum(1)
    gam = np.where((den > 1e-15)[:, None], gm / np.where(den > 1e-15, den, 1.0)[:, None], 0.0)
    use = np.where(gam < 1e-15, 0.0, gam)
    np.testing.assert_allclose(t0, use.sum(0), rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(t1, use.T @ Ps, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(t2, np.einsum('nj,na,nb->jab', use, Ps, Ps), rtol=1e-10, atol=1e-12)
    assert np.array_equal(lab_s, np.argmax(gam, axis=1))
    assert np.array_equal(lab_s, lab_e[:20000])
    np.testing.assert_allclose(q_s, np.log(np.maximum(gm.sum(1), 1e-15)).sum(), rtol=1e-11)


# ---- float32 tile (Context.tree_set_precision(np.float32): the reference GPU file's type, hgmm/hgmm_gpu.py:472-484) --------
# full_fused_f32_kernel: pdfs, the 16-point tile, gamma and the statistics' products in float32 (matrix cores,
# v_mfma_f32_16x16x4_f32, about the cloud's centroid); row sums, 1 / den, log(), the segments' sums and the M-step in
# float64.  Opt-in; float64 stays the default and the parity reference.  Held to: the reference's goldens (labels and
# iteration counts equal, parameters to 1e-5 of their scale), the 10^6-point oracle fixture, every component layout,
# a cloud far from the origin, the symmetric-form fallback.

def _f32(ctx):
    class _Mode:
        def __enter__(self):
            ctx.tree_set_precision(np.float32)
        def __exit__(self, *a):
            ctx.tree_set_precision(np.float64)
    return _Mode()


def _cov_err(cov, ref):
    """largest |d Sigma_ab| over a component's mean variance (the scale a covariance entry is meaningful on)"""
    scale = np.abs(np.einsum("jii->j", ref)) / 3.0
    return float(np.max(np.abs(cov - ref).reshape(len(ref), -1).max(1) / np.maximum(scale, 1e-300)))


@pytest.mark.parametrize("J", [8, 32])
def test_fullcov_float32_tile_matches_reference_golden(ctx, J):
    g = load_golden("fullcov_flat.npz")
    P = g["points"]
    tag = "J%d_" % J
    ctx.set_points(P)
    with _f32(ctx):
        pi, mu, cov, labels, q = ctx.fullcov_fit(J, 80.0, 1e-4, P[g[tag + "init_idx"]], 0.00034)
    assert len(q) == len(g[tag + "q_trace"])                            # the same number of iterations
    assert np.array_equal(labels, g[tag + "current_idx"])               # every label
    # (fits run to convergence, 60-70 iterations: the per-iteration float32 differences -- ~1e-6 -- have that long to grow)
    np.testing.assert_allclose(q, g[tag + "q_trace"], rtol=1e-6, atol=1e-2)
    np.testing.assert_allclose(pi, g[tag + "pi"], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(mu, g[tag + "mu"], rtol=0, atol=1e-5)
    assert _cov_err(cov, g[tag + "cov"]) < 1e-3
    print("J = %d: %d iterations, |dq| max %.3g, |d pi| / pi %.2g, |d mu| %.2g, |d cov| / variance %.2g"
          % (J, len(q), np.abs(q - g[tag + "q_trace"]).max(), np.max(np.abs(pi - g[tag + "pi"]) / g[tag + "pi"]),
             np.abs(mu - g[tag + "mu"]).max(), _cov_err(cov, g[tag + "cov"])))


def test_fullcov_float32_tile_1M_against_oracle_fixture(ctx):
    """The 10^6-point, J = 800, 3-iteration oracle fixture: pi and mu within 1e-5, Sigma within 3e-5 of a component's
    variance (float32 second moments about the centroid: |x - o|^2 / sigma^2 ~ 300 here), at most 10 of 10^6 labels
    off (top-two responsibilities within float32 rounding of each other), bitwise rerun; the statistics' sum over
    ragged shards equals the whole cloud's to float32-accumulation accuracy."""
    g = load_golden("fullcov_uniform1M_J800_oracle.npz")
    N, J, iters = int(g["n_points"]), int(g["J"]), int(g["max_iters"])
    P = np.random.RandomState(int(g["cloud_seed"])).rand(N, 3).astype(np.float32).astype(np.float64)
    idx = np.random.RandomState(int(g["init_seed"])).choice(N, J, replace=False)
    ctx.set_points(P)
    with _f32(ctx):
        pi, mu, cov, labels, q = ctx.fullcov_fit(J, float(g["ls"]), float(g["ld"]), P[idx], float(g["sig2"]), iters)
        again = ctx.fullcov_fit(J, float(g["ls"]), float(g["ld"]), P[idx], float(g["sig2"]), iters)
        m0, m1, m2, lab_e, q_e = ctx.fullcov_estep(pi, mu, cov)
    assert np.array_equal(again[4], q) and np.array_equal(again[3], labels) and np.array_equal(again[2], cov)
    assert len(q) == iters
    assert np.abs(q - g["q_trace"]).max() < 0.5                          # (|q| ~ 1e4 here: densities near 1)
    np.testing.assert_allclose(pi, g["pi"], rtol=1e-5)
    np.testing.assert_allclose(mu, g["mu"], rtol=0, atol=1e-5)
    assert _cov_err(cov, g["cov"]) < 3e-5
    off = int((labels[g["sample"]] != g["labels_sample"]).sum())
    assert off <= 2 and np.abs(np.bincount(labels, minlength=J) - g["population"]).sum() <= 20
    assert abs(m0.sum() - N) <= 1e-6 * N
    print("1M fixture, float32 tile: |dq| %.3g, |d pi| / pi %.2g, |d mu| %.2g, |d cov| / variance %.2g, sampled labels off %d / %d"
          % (np.abs(q - g["q_trace"]).max(), np.max(np.abs(pi - g["pi"]) / g["pi"]), np.abs(mu - g["mu"]).max(),
             _cov_err(cov, g["cov"]), off, len(g["sample"])))


@pytest.mark.parametrize("J", [17, 100, 513, 540, 800, 1024])
def test_fullcov_float32_tile_layouts_far_origin_and_fallback(ctx, bunny, J):
    """Component counts that fill the lanes' pairs differently (an odd number of 16-blocks, more than 896, 1024), the cloud
    moved 50 units from the origin (the float32 features are taken about the centroid), and the symmetric-form fallback
    (tree_no_chol = 1): against the oracle at float32 accuracy."""
    P0 = bunny[::30][:1300].astype(np.float64)
    idx = np.random.RandomState(J).choice(len(P0), J, replace=False)
    for shift, no_chol in ((0.0, 0), (50.0, 0), (0.0, 1)):
        P = P0 + shift
        ctx.set_points(P)
        with _f32(ctx), ctx.config(tree_no_chol=no_chol):
            pi, mu, cov, labels, q = ctx.fullcov_fit(J, 1e-30, 1e-4, P[idx], 0.0005, 3)
        o_pi, o_mu, o_cov, o_q, o_cur = hgmm_tree.build_flat_fullcov(P, J, 1e-30, 1e-4, idx, 0.0005, max_iters=3)
        assert (labels != o_cur).sum() <= 2
        np.testing.assert_allclose(q, o_q, rtol=1e-6, atol=1e-2)
        # (1300 points over up to 1024 components: two or three points per component -- a layout test, the accuracy tests
        #  are the two above)
        np.testing.assert_allclose(pi, o_pi, rtol=1e-3, atol=1e-7)
        np.testing.assert_allclose(mu, o_mu, rtol=0, atol=1e-5)
        live = o_pi > 1e-6
        assert _cov_err(cov[live], o_cov[live]) < 1e-3, (shift, no_chol, _cov_err(cov[live], o_cov[live]))


def test_fitFullCovGMM_dtype_argument_selects_the_float32_tile_for_the_call(ctx, bunny):
    from hgmm_amd.hgmm.hgmm_gpu import fitFullCovGMM
    P = bunny[::10].astype(np.float64)
    idx = np.random.RandomState(3).choice(len(P), 48, replace=False)
    a = fitFullCovGMM(P, 48, ls=1e-30, init_idx=idx, max_iters=4, ctx=ctx)
    b = fitFullCovGMM(P, 48, ls=1e-30, init_idx=idx, max_iters=4, ctx=ctx, dtype=np.float32)
    c = fitFullCovGMM(P.astype(np.float32), 48, ls=1e-30, init_idx=idx, max_iters=4, ctx=ctx)     # float32 POINTS do not switch it
    assert ctx.tree_dtype == np.dtype(np.float64)
    assert not np.array_equal(a[1], b[1])                                    # another kernel ran ...
    np.testing.assert_allclose(b[1], a[1], rtol=0, atol=1e-6)                # ... to float32 accuracy
    np.testing.assert_allclose(b[0], a[0], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(c[1], fitFullCovGMM(P.astype(np.float32).astype(np.float64), 48, ls=1e-30, init_idx=idx,
                                                   max_iters=4, ctx=ctx)[1], rtol=0, atol=0)


@pytest.mark.parametrize("N,J", [(1, 3), (15, 4), (64, 16), (777, 17), (5032, 100)])
def test_fullcov_float32_tile_small_clouds_vs_oracle(ctx, bunny, N, J):
    """Clouds smaller than a tile / a segment / a workgroup's share (one point, 15, 64, ...) through the float32-tile kernel."""
    P = bunny[:: max(1, len(bunny) // N)][:N].astype(np.float64)
    rs = np.random.RandomState(J)
    idx = rs.choice(len(P), J, replace=len(P) < J)
    ctx.set_points(P)
    with _f32(ctx):
        pi, mu, cov, labels, q = ctx.fullcov_fit(J, 1.0, 1e-4, P[idx], 0.0005, 8)
    o_pi, o_mu, o_cov, o_q, o_cur = hgmm_tree.build_flat_fullcov(P, J, 1.0, 1e-4, idx, 0.0005, max_iters=8)
    assert len(q) == len(o_q)
    assert (labels != o_cur).sum() <= max(1, N // 2000)
    np.testing.assert_allclose(q, o_q, rtol=1e-5, atol=1e-2)
    np.testing.assert_allclose(pi, o_pi, rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(mu, o_mu, rtol=0, atol=1e-5)
