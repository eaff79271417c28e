"""GPU parity tests of the BATCHED hierarchical-GMM path (hgmm_tree_build_batch / hgmm_tree_register_batch: B independent
scan pairs per launch set) through the C ABI.

The reference's unit of work is one pair, registration_gmmtree(source, target) (src/python/hgmm/hgmm_gpu.py:802-807 =
buildGMMTree 466-548 + GMMTree.registration 754-768).  The bar for the batched entries is the strictest the domain
offers: every pair's tree, per-level iteration counts, q trace and recovered (R, t) are BITWISE what the serial entries
give for that pair alone -- which are themselves held to the reference's goldens / the oracle by tests/test_tree_gpu.py.
The goldens are checked here once more directly (tolerances as there)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from oracle import hgmm_tree

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import hgmm_amd
    c = hgmm_amd.Context(0)
    yield c
    c.close()


def serial_build(ctx, P, L, ls, ld, idx, sig2, max_iters=1000):
    P = np.ascontiguousarray(P, dtype=np.float64)
    ctx.set_points(P)
    return ctx.tree_build(L, ls, ld, P[idx], sig2, max_iters, want_leaf=False)


def batch_build(ctx, clouds, L, ls, ld, idx, sig2, max_iters=1000):
    arrs = ctx.set_points_batch(clouds)
    init = np.stack([a[idx] for a in arrs])
    return ctx.tree_build_batch([len(a) for a in arrs], L, ls, ld, init, sig2, max_iters, want_trace=True)


def ragged_clouds(bunny):
    """Clouds of very different sizes and shapes: scans, sub-samples, blobs, one smaller than a 256-point chunk."""
    rs = np.random.RandomState(5)
    b = bunny.astype(np.float64)
    b45 = np.load(os.path.join(GOLDEN, "bun045_xyz.npy")).astype(np.float64)
    blobs = (rs.rand(6, 3)[rs.randint(6, size=3100)] * 0.2 + 0.01 * rs.randn(3100, 3))
    return [b[::20], b45[::3], blobs, b[::7], b[5:205] * 1.0, b[::2], load_golden("hgmm_build_L3.npz")["points"]]


@pytest.mark.parametrize("L,ls,sig2", [(1, 20.0, 0.004), (2, 20.0, 0.004), (3, 20.0, 0.004), (3, 80.0, 0.00034), (4, 20.0, 0.004)])
def test_build_batch_is_bitwise_the_serial_build(ctx, bunny, L, ls, sig2):
    clouds = ragged_clouds(bunny)
    T = hgmm_tree.n_total(L)
    idx = np.random.RandomState(72).randint(T, size=T)
    idx = np.minimum(idx, min(len(c) for c in clouds) - 1)            # (the smallest cloud has 200 points)
    (pi, mu, cov), iters, traces = batch_build(ctx, clouds, L, ls, 1e-4, idx, sig2)
    for b, P in enumerate(clouds):
        s_pi, s_mu, s_cov, _, s_iters, s_q = serial_build(ctx, P, L, ls, 1e-4, idx, sig2)
        assert list(iters[b]) == list(s_iters), (b, iters[b], s_iters)
        assert np.array_equal(traces[b], s_q), (b, np.abs(traces[b] - s_q).max())
        assert np.array_equal(pi[b], s_pi) and np.array_equal(mu[b], s_mu) and np.array_equal(cov[b], s_cov), b
    print("L=%d: per-level iterations of the %d clouds" % (L, len(clouds)), iters.tolist())
    assert len({tuple(r) for r in iters.tolist()}) > 1                 # the clouds do stop at different iterations


@pytest.mark.parametrize("L,ls,sig2", [(1, 20.0, 0.004), (2, 20.0, 0.004), (3, 20.0, 0.004), (3, 80.0, 0.00034), (4, 20.0, 0.004)])
def test_build_batch_float32_pdfs_is_bitwise_the_serial_build_and_keeps_the_float64_trees(ctx, bunny, L, ls, sig2):
    """hgmm_tree_set_precision(F32_PDF) on small clouds and forests (round 6): the stop rule's log-likelihood in float32.
    (i) batch == serial bit for bit in that mode too (trees, iteration counts, q traces); (ii) against the float64 mode:
    |dq| stays far below the stop threshold, and where a level stops after the same number of iterations the tree is
    the float64 tree bit for bit (E-step and moments are float64 in both) -- here: on every cloud."""
    clouds = ragged_clouds(bunny)
    T = hgmm_tree.n_total(L)
    idx = np.random.RandomState(72).randint(T, size=T)
    idx = np.minimum(idx, min(len(c) for c in clouds) - 1)
    (pi64, mu64, cov64), iters64, traces64 = batch_build(ctx, clouds, L, ls, 1e-4, idx, sig2)
    ctx.tree_set_precision(np.float32)
    try:
        (pi, mu, cov), iters, traces = batch_build(ctx, clouds, L, ls, 1e-4, idx, sig2)
        worst = 0.0
        for b, P in enumerate(clouds):
            s_pi, s_mu, s_cov, _, s_iters, s_q = serial_build(ctx, P, L, ls, 1e-4, idx, sig2)
            assert list(iters[b]) == list(s_iters), (b, iters[b], s_iters)
            assert np.array_equal(traces[b], s_q), (b, np.abs(traces[b] - s_q).max())
            assert np.array_equal(pi[b], s_pi) and np.array_equal(mu[b], s_mu) and np.array_equal(cov[b], s_cov), b
            assert list(iters[b]) == list(iters64[b]), (b, iters[b], iters64[b])
            assert np.array_equal(pi[b], pi64[b]) and np.array_equal(mu[b], mu64[b]) and np.array_equal(cov[b], cov64[b]), b
            worst = max(worst, np.abs(traces[b] - traces64[b]).max())
    finally:
        ctx.tree_set_precision(np.float64)
    print("L=%d: largest |q_f32 - q_f64| over %d clouds' traces: %.3g (stop threshold %g)" % (L, len(clouds), worst, ls))
    assert worst < 0.01 * ls
    assert (worst > 0.0) == (L > 1)                # (level 0 has no log-likelihood kernel in either mode; below it the float32 one ran)


@pytest.mark.parametrize("name", ["hgmm_build_L2.npz", "hgmm_build_L3.npz"])
def test_build_batch_matches_reference_golden(ctx, bunny, name):
    """The reference's own build goldens as members of a batch (beside other clouds)."""
    g = load_golden(name)
    P, L = g["points"], int(g["L"])
    clouds = [bunny[::9].astype(np.float64), P, bunny[::33].astype(np.float64), P[::-1].copy()]
    (pi, mu, cov), iters, traces = batch_build(ctx, clouds, L, float(g["ls"]), float(g["ld"]), g["init_idx"], float(g["sig2"]))
    assert list(iters[1]) == list(g["iters_per_level"])
    np.testing.assert_allclose(traces[1], g["q_trace"], rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(pi[1], g["pi"], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(mu[1], g["mu"], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(cov[1], g["cov"], rtol=1e-7, atol=1e-14)


def test_build_batch_iteration_budget_and_single_cloud(ctx, bunny):
    """Levels that end on the iteration budget (the close launch instead of a follower) and B = 1."""
    clouds = [bunny[::11].astype(np.float64), bunny[::5].astype(np.float64)]
    L = 3
    T = hgmm_tree.n_total(L)
    idx = np.random.RandomState(72).randint(T, size=T)
    for budget in (1, 2, 5):
        (pi, mu, cov), iters, traces = batch_build(ctx, clouds, L, 1e-30, 1e-4, idx, 0.004, max_iters=budget)
        assert (iters == budget).all()
        for b, P in enumerate(clouds):
            s = serial_build(ctx, P, L, 1e-30, 1e-4, idx, 0.004, max_iters=budget)
            assert np.array_equal(pi[b], s[0]) and np.array_equal(mu[b], s[1]) and np.array_equal(cov[b], s[2])
            assert np.array_equal(traces[b], s[5])
    (pi, mu, cov), iters, traces = batch_build(ctx, clouds[:1], L, 20.0, 1e-4, idx, 0.004)
    s = serial_build(ctx, clouds[0], L, 20.0, 1e-4, idx, 0.004)
    assert np.array_equal(pi[0], s[0]) and np.array_equal(cov[0], s[2]) and list(iters[0]) == list(s[4])


def _moved(P, deg, axis, shift, rs=None, noise=0.0):
    axis = np.asarray(axis, dtype=np.float64) / np.linalg.norm(axis)
    th = np.deg2rad(deg)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    Q = P @ R.T + np.asarray(shift)
    if rs is not None and noise:
        Q = Q + noise * rs.randn(*Q.shape)
    return Q


def test_registration_batch_is_bitwise_the_serial_registration(ctx, bunny):
    """registration_gmmtree_batch == [registration_gmmtree(s, t) ...]: transformations, q and iteration counts, bit for
    bit, on ragged pairs (different sources, targets of other lengths than their sources, different motions)."""
    from hgmm_amd.hgmm.hgmm_gpu import GMMTree, registration_gmmtree_batch
    rs = np.random.RandomState(11)
    b = bunny.astype(np.float64)
    b45 = np.load(os.path.join(GOLDEN, "bun045_xyz.npy")).astype(np.float64)
    srcs = [b[::4], b[::9], b45[::5], b[1::6], b[::13]]
    pairs = [(srcs[0], _moved(srcs[0][::2], 7.0, [0.2, 1, 0.1], [0.004, -0.002, 0.003], rs, 2e-4)),
             (srcs[1], _moved(srcs[1], 3.0, [1, 0.3, 0.0], [0.001, 0.0, -0.002])),
             (srcs[2], _moved(b45[2::7], 10.0, [0, 0.2, 1], [-0.003, 0.004, 0.0], rs, 1e-4)),
             (srcs[3], _moved(srcs[3], 0.0, [0, 0, 1], [0.0, 0.0, 0.0])),                 # identity: stops at once
             (srcs[4], _moved(b[::3], 5.0, [1, 1, 1], [0.002, 0.002, 0.002]))]
    kw = dict(tree_level=3, lambda_c=0.01, ls=20, sig2=0.004)
    res, info = registration_gmmtree_batch(pairs, maxiter=20, tol=1e-4, ctx=ctx, return_info=True, **kw)
    for k, (s, t) in enumerate(pairs):
        gt = GMMTree(s, ctx=ctx, **kw)
        ref = gt.registration(t, 20, 1e-4)
        assert int(gt.n_iter_) == info["registration_iters"][k], (k, gt.n_iter_, info["registration_iters"])
        assert np.array_equal(ref.transformation.rot, res[k].transformation.rot), k
        assert np.array_equal(ref.transformation.t, res[k].transformation.t), k
        assert np.array_equal(np.ravel(ref.q), np.ravel(res[k].q)), k
    print("registration iterations per pair:", info["registration_iters"], "build:", info["build_iters"].tolist())
    assert len(set(info["registration_iters"])) > 1


def test_registration_batch_of_float32_scans_is_bitwise_the_serial_registration(ctx, bunny):
    """float32 scans (the reference's GPU file casts to float32, hgmm_gpu.py:472): both mirrors take the stop rule's pdfs
    in float32 for them -- the serial GMMTree and the batch still agree bit for bit; a batch that mixes float32 and
    float64 sources runs as two batches and returns the pairs in the caller's order."""
    from hgmm_amd.hgmm.hgmm_gpu import GMMTree, registration_gmmtree_batch
    b32 = bunny.astype(np.float32)
    b45 = np.load(os.path.join(GOLDEN, "bun045_xyz.npy")).astype(np.float32)
    srcs = [b32[::4], b32[::9].astype(np.float64), b45[::5], b32[1::6]]
    pairs = [(srcs[0], _moved(srcs[0][::2].astype(np.float64), 7.0, [0.2, 1, 0.1], [0.004, -0.002, 0.003])),
             (srcs[1], _moved(srcs[1], 3.0, [1, 0.3, 0.0], [0.001, 0.0, -0.002])),
             (srcs[2], _moved(b45[2::7].astype(np.float64), 10.0, [0, 0.2, 1], [-0.003, 0.004, 0.0]).astype(np.float32)),
             (srcs[3], _moved(srcs[3].astype(np.float64), 5.0, [1, 1, 1], [0.002, 0.002, 0.002]))]
    kw = dict(tree_level=3, lambda_c=0.01, ls=20, sig2=0.004)
    res, info = registration_gmmtree_batch(pairs, maxiter=20, tol=1e-4, ctx=ctx, return_info=True, **kw)
    assert ctx.tree_dtype == np.dtype(np.float64)                      # the context's own setting came back
    for k, (s, t) in enumerate(pairs):
        gt = GMMTree(s, ctx=ctx, **kw)
        ref = gt.registration(t, 20, 1e-4)
        assert int(gt.n_iter_) == info["registration_iters"][k], (k, gt.n_iter_, info["registration_iters"])
        assert np.array_equal(ref.transformation.rot, res[k].transformation.rot), k
        assert np.array_equal(ref.transformation.t, res[k].transformation.t), k
        assert np.array_equal(np.ravel(ref.q), np.ravel(res[k].q)), k
    assert info["build_iters"].shape == (4, 3)


def test_registration_batch_matches_reference_trace(ctx):
    """The reference's recorded registration (hgmm_reg_L2.npz: per-iteration (R, t) of its CPU twin) through the batched
    entries: tree uploaded per pair is not part of this path, so the golden's cloud is built here and only the final
    transform of the serial mirror is compared bitwise; the per-iteration trace goes against the serial trace."""
    g = load_golden("hgmm_reg_L2.npz")
    P = g["points"]
    L = int(g["L"])
    T = hgmm_tree.n_total(L)
    idx = np.random.RandomState(72).randint(T, size=T)
    targets = [g["rot10_target"], g["rot30_target"], g["rot10_target"][::2]]
    arrs = ctx.set_points_batch([P] * len(targets))
    ctx.tree_build_batch([len(P)] * len(targets), L, 20.0, 1e-4, np.stack([a[idx] for a in arrs]), 0.004, want_tables=False)
    ctx.tree_set_targets_batch(targets)
    B = len(targets)
    rot, t, iters, q, status, traces = ctx.tree_register_batch(np.tile(np.eye(3), (B, 1, 1)), np.zeros((B, 3)), 1.0,
                                                               float(g["lambda_c"]), 8, 1e-9, want_trace=True)
    for k, tg in enumerate(targets):
        pi, mu, cov, _, _, _ = serial_build(ctx, P, L, 20.0, 1e-4, idx, 0.004)
        ctx.tree_set_nodes(L, pi, mu, cov)
        ctx.tree_set_target(tg)
        s_rot, s_t, s_done, s_q, s_status, s_trace = ctx.tree_register(np.eye(3), np.zeros(3), 1.0, float(g["lambda_c"]), 8,
                                                                       1e-9, None, want_trace=True)
        assert s_done == iters[k] and s_status == status[k]
        assert np.array_equal(s_trace, traces[k]), k
        assert np.array_equal(s_rot, rot[k]) and np.array_equal(s_t, t[k]) and s_q == q[k]


def test_registration_batch_ill_conditioned_pair_falls_back_like_the_serial_path(ctx, bunny):
    """A one-level tree and a target of ONE point make the 6 x 6 normal equations singular (one node with mass: rank 3): the serial loop hands that iteration to the
    host's stacked least squares (status 2); in a batch the pair leaves the batch and is finished the same way, while the
    other pairs are not disturbed."""
    from hgmm_amd.hgmm.hgmm_gpu import GMMTree, registration_gmmtree_batch
    b = bunny.astype(np.float64)
    src = b[::10]
    pairs = [(src, _moved(src, 4.0, [0, 1, 0], [0.001, 0.001, 0.0])), (src, src[:1] + 0.001),
             (b[::12], _moved(b[::12], 6.0, [1, 0, 0], [0.0, 0.002, 0.0]))]
    kw = dict(tree_level=1, lambda_c=0.01, ls=20, sig2=0.004)
    res, info = registration_gmmtree_batch(pairs, maxiter=6, tol=1e-6, ctx=ctx, return_info=True, **kw)
    assert info["status"][1] == 2
    for k, (s, t) in enumerate(pairs):
        gt = GMMTree(s, ctx=ctx, **kw)
        ref = gt.registration(t, 6, 1e-6)
        assert np.array_equal(ref.transformation.rot, res[k].transformation.rot), k
        assert np.array_equal(ref.transformation.t, res[k].transformation.t), k
        assert int(gt.n_iter_) == info["registration_iters"][k]


def test_batch_rejects_what_it_cannot_reproduce(ctx, bunny):
    import hgmm_amd
    P = bunny.astype(np.float64)
    ctx.set_points(P)
    T = hgmm_tree.n_total(2)
    with pytest.raises(hgmm_amd.HgmmError):                            # counts do not add up to the resident cloud
        ctx.tree_build_batch([100, 200], 2, 20.0, 1e-4, np.zeros((2, T, 3)), 0.004)
    with pytest.raises(hgmm_amd.HgmmError):                            # no forest of that size resident
        ctx.tree_register_batch(np.tile(np.eye(3), (7, 1, 1)), np.zeros((7, 3)))


def test_solve_on_device_through_the_mirrors(ctx, bunny):
    """``GMMTree(..., solve_on_device=True)`` / ``registration_gmmtree_batch(..., solve_on_device=True)``: the per-context
    option reg_device_solve for the duration of the call -- serial and batched still bit for bit each other, within 1e-9
    of the host-solve path, the option back to what it was."""
    from hgmm_amd.hgmm.hgmm_gpu import GMMTree, registration_gmmtree_batch
    b = bunny.astype(np.float64)
    srcs = [b[::5], b[::8], b[2::9]]
    pairs = [(s, _moved(s, 4.0 + k, [0.3, 1, 0.2 * k], [0.002, -0.001 * k, 0.001])) for k, s in enumerate(srcs)]
    kw = dict(tree_level=3, lambda_c=0.01, ls=20, sig2=0.004)
    res, info = registration_gmmtree_batch(pairs, maxiter=20, tol=1e-4, ctx=ctx, return_info=True, solve_on_device=True, **kw)
    assert ctx.config_get("reg_device_solve") == 0
    for k, (s, t) in enumerate(pairs):
        gt = GMMTree(s, ctx=ctx, solve_on_device=True, **kw)
        dev = gt.registration(t, 20, 1e-4)
        assert int(gt.n_iter_) == info["registration_iters"][k]
        assert np.array_equal(dev.transformation.rot, res[k].transformation.rot) and np.array_equal(dev.transformation.t, res[k].transformation.t)
        host = GMMTree(s, ctx=ctx, **kw).registration(t, 20, 1e-4)
        np.testing.assert_allclose(dev.transformation.rot, host.transformation.rot, rtol=0, atol=1e-9)
        np.testing.assert_allclose(dev.transformation.t, host.transformation.t, rtol=0, atol=1e-9)
    assert ctx.config_get("reg_device_solve") == 0


def test_float32_clouds_are_widened_on_the_device_exactly(ctx, bunny):
    """hgmm_set_points_batch_f32 / hgmm_tree_set_targets_batch_f32: float32 rows in, widened to float64 on the device --
    trees and transformations bit for bit those of the float64 entries on the host-widened arrays."""
    b32 = bunny.astype(np.float32)
    clouds32 = [b32[::6], b32[1::9], b32[::14]]
    clouds64 = [c.astype(np.float64) for c in clouds32]
    T = hgmm_tree.n_total(3)
    idx = np.random.RandomState(72).randint(T, size=T)
    idx = np.minimum(idx, min(len(c) for c in clouds32) - 1)
    out = {}
    for name, clouds in (("f32", clouds32), ("f64", clouds64)):
        arrs = ctx.set_points_batch(clouds)
        assert arrs[0].dtype == (np.float32 if name == "f32" else np.float64)
        tabs, iters, _ = ctx.tree_build_batch([len(a) for a in arrs], 3, 20.0, 1e-4, np.stack([a[idx] for a in arrs]), 0.004)
        tg = [_moved(c.astype(np.float64), 5.0, [0.1, 1, 0.3], [0.003, 0.0, -0.002]).astype(c.dtype) for c in clouds]
        ctx.tree_set_targets_batch(tg)
        reg = ctx.tree_register_batch(np.tile(np.eye(3), (3, 1, 1)), np.zeros((3, 3)), 1.0, 0.01, 12, 1e-6)
        out[name] = (tabs, iters, reg)
    (ta, ia, ra), (tb, ib, rb) = out["f32"], out["f64"]
    assert np.array_equal(ia, ib)
    for x, y in zip(ta, tb):
        assert np.array_equal(x, y)
    # (the float64 targets were made from the float32 ones' values: _moved(...).astype(float32) widened again)
    tg64 = [_moved(c.astype(np.float64), 5.0, [0.1, 1, 0.3], [0.003, 0.0, -0.002]).astype(np.float32).astype(np.float64) for c in clouds32]
    ctx.set_points_batch(clouds64)
    ctx.tree_build_batch([len(a) for a in clouds64], 3, 20.0, 1e-4, np.stack([a[idx] for a in clouds64]), 0.004, want_tables=False)
    ctx.tree_set_targets_batch(tg64)
    rc = ctx.tree_register_batch(np.tile(np.eye(3), (3, 1, 1)), np.zeros((3, 3)), 1.0, 0.01, 12, 1e-6)
    assert np.array_equal(ra[0], rc[0]) and np.array_equal(ra[1], rc[1]) and np.array_equal(ra[2], rc[2])


def test_a_pair_too_large_for_the_batch_runs_serially_inside_the_batched_call(ctx, bunny):
    """Sources of >= 400 000 points are refused by hgmm_tree_build_batch (they fill the chip alone and take another
    log-likelihood kernel); registration_gmmtree_batch runs those pairs through the serial call and batches the rest."""
    import hgmm_amd
    from hgmm_amd.hgmm.hgmm_gpu import GMMTree, registration_gmmtree_batch
    rs = np.random.RandomState(2)
    cen = rs.rand(40, 3) * 0.2
    big = cen[rs.randint(40, size=410000)] + 0.004 * rs.randn(410000, 3)
    small = bunny[::10].astype(np.float64)
    pairs = [(small, _moved(small, 5.0, [0, 1, 0], [0.002, 0.0, 0.001])), (big, _moved(big[::50], 2.0, [1, 0, 0], [0.001, 0.001, 0.0])),
             (small[::2], _moved(small[::2], 3.0, [0, 0, 1], [0.0, 0.001, 0.0]))]
    kw = dict(tree_level=2, lambda_c=0.01, ls=80, sig2=0.004)
    with pytest.raises(hgmm_amd.HgmmError):
        ctx.set_points_batch([p[0] for p in pairs])
        ctx.tree_build_batch([len(p[0]) for p in pairs], 2, 80.0, 1e-4, np.zeros((3, hgmm_tree.n_total(2), 3)), 0.004)
    res, info = registration_gmmtree_batch(pairs, maxiter=10, tol=1e-4, ctx=ctx, return_info=True, **kw)
    assert list(info["build_iters"][1]) == [-1, -1] and (info["build_iters"][[0, 2]] > 0).all()
    for k, (s, t) in enumerate(pairs):
        gt = GMMTree(s, ctx=ctx, **kw)
        ref = gt.registration(t, 10, 1e-4)
        assert np.array_equal(ref.transformation.rot, res[k].transformation.rot) and np.array_equal(ref.transformation.t, res[k].transformation.t), k
        assert int(gt.n_iter_) == info["registration_iters"][k]


def test_the_bun_conf_scan_pair_in_a_batch_is_bitwise_the_serial_call():
    """The pair bench.py --mode pairs times -- bun000.ply against bun045.ply placed by the reference's bun.conf and moved by
    a known rigid motion -- as members of a batch (float32 and float64 scans): transformation, q and iteration counts bit
    for bit the serial registration_gmmtree's, and within the bench's 6 mm of the ground truth."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import hgmm_amd
    from hgmm_amd.hgmm.hgmm_gpu import GMMTree, registration_gmmtree_batch
    c = hgmm_amd.Context(0)
    try:
        source, pairs = bench.scan_pairs(0, 3)
        for dt in (np.float32, np.float64):
            src = source.astype(dt)
            batch = [(src, t.astype(dt)) for t, _ in pairs]
            res, info = registration_gmmtree_batch(batch, maxiter=bench.PAIR_MAXITER, tol=bench.PAIR_TOL, ctx=c, return_info=True,
                                                   **bench.PAIR_KW)
            for k, (s, t) in enumerate(batch):
                gt = GMMTree(s, ctx=c, **bench.PAIR_KW)
                ref = gt.registration(t, bench.PAIR_MAXITER, bench.PAIR_TOL)
                assert np.array_equal(ref.transformation.rot, res[k].transformation.rot), (dt, k)
                assert np.array_equal(ref.transformation.t, res[k].transformation.t), (dt, k)
                assert int(gt.n_iter_) == info["registration_iters"][k]
                err = np.linalg.norm(res[k].transformation.transform(source) - pairs[k][1], axis=1).mean()
                assert err < 0.006, (dt, k, err)
    finally:
        c.close()
