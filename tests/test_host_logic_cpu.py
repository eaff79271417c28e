"""CPU tests of the host-side logic that stays in NumPy in the product (registration M-step,
transform algebra, index algebra) against the oracle / golden vectors.  No GPU needed."""
import numpy as np

from conftest import load_golden
from oracle import hgmm_tree


def test_tree_index_algebra():
    from hgmm_amd.hgmm import hgmm_gpu as H
    for L in range(1, 6):
        assert H.n_total_nodes(L) == hgmm_tree.n_total(L)
    assert [int(H.level(l)) for l in range(5)] == [hgmm_tree.level(l) for l in range(5)] == [0, 8, 72, 584, 4680]
    assert H.child2(-1) == 0 and H.child2(3) == 32


def test_registration_mstep_matches_oracle_and_reference():
    """GMMTree.maximization_step (vectorised NumPy) == the reference's per-node loop: feed it the
    reference's own E-step moments and compare with the reference's first-iteration transform."""
    from hgmm_amd.hgmm.hgmm_gpu import GMMTree, RigidTransformation, EstepResult
    g = load_golden("hgmm_reg_L2.npz")
    for deg in (10, 30):
        tag = "rot%d_" % deg
        gt = GMMTree(None, tree_level=int(g["L"]), lambda_c=float(g["lambda_c"]))
        gt.set_nodes(g["pi"], g["mu"], g["cov"])
        res = gt.maximization_step(EstepResult(g[tag + "m0"], g[tag + "m1"], g[tag + "m2"]), RigidTransformation())
        rot, t, q = hgmm_tree.reg_m_step(g[tag + "m0"], g[tag + "m1"], g[tag + "m2"], g["mu"], g["cov"],
                                         np.identity(3), np.zeros(3))
        np.testing.assert_allclose(res.transformation.rot, rot, atol=1e-12)
        np.testing.assert_allclose(res.transformation.t, t, atol=1e-12)
        np.testing.assert_allclose(res.q, q, rtol=1e-10)          # QR residual == lstsq residual
        inv = res.transformation.inverse()
        np.testing.assert_allclose(inv.rot, g[tag + "iter_rot"][0], atol=1e-9)
        np.testing.assert_allclose(inv.t, g[tag + "iter_t"][0], atol=1e-9)


def test_rigid_transformation_roundtrip():
    from hgmm_amd.hgmm.hgmm_gpu import RigidTransformation, twist_trans
    rs = np.random.RandomState(0)
    R, t = twist_trans(np.array([0.1, -0.2, 0.3, 1.0, 2.0, 3.0]))
    np.testing.assert_allclose(R @ R.T, np.identity(3), atol=1e-14)
    tf = RigidTransformation(R, t, 1.0)
    X = rs.rand(10, 3)
    np.testing.assert_allclose(tf.inverse().transform(tf.transform(X)), X, atol=1e-14)


def test_noisy_target_generator(bunny):
    from hgmm_amd.hgmm.hgmm_gpu import prepare_source_and_target_rigid_3d, euler_matrix_xyz
    R = euler_matrix_xyz(0.1, -0.2, 0.3)
    np.testing.assert_allclose(R @ R.T, np.identity(3), atol=1e-14)
    assert abs(np.linalg.det(R) - 1) < 1e-14
    np.testing.assert_allclose(euler_matrix_xyz(0, 0, np.pi / 2) @ [1, 0, 0], [0, 1, 0], atol=1e-15)
    rs = np.random.RandomState(0)
    src, tgt = prepare_source_and_target_rigid_3d(bunny.astype(np.float64), n_random=100, rng=rs)
    assert tgt.shape == (len(src) + 100, 3)
    th = np.deg2rad(30.0)
    Rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    back = tgt[:len(src)] @ Rz                      # undo the rotation: inliers sit on the source
    d = np.abs(back[:, None, :] - src[None, :500, :]).sum(2).min(0)
    assert np.median(d) < 0.01


def test_batch_mirror_groups_pairs_by_the_sources_type_and_keeps_the_callers_order():
    """registration_gmmtree_batch (host logic only, a recording stand-in for the context): pairs whose SOURCE is float32 run
    as one batch with the float32-pdf stop rule, the others as another with float64; results and per-pair information
    come back in the caller's order; the context's own precision setting is restored."""
    from hgmm_amd.hgmm import hgmm_gpu as H

    class Recorder:
        def __init__(self):
            self.tree_dtype = np.dtype(np.float64)
            self.calls = []
            self._n = 0

        def tree_set_precision(self, dt):
            self.tree_dtype = np.dtype(dt)
            self.calls.append(("precision", str(np.dtype(dt))))

        def set_points_batch(self, clouds):
            self._n = len(clouds)
            self.calls.append(("sources", [len(c) for c in clouds]))
            return clouds

        def tree_build_batch(self, counts, L, ls, ld, init_mu, sig2, want_tables=False):
            self.calls.append(("build", str(self.tree_dtype), list(counts)))
            return (None, None, None), np.tile(np.arange(1, L + 1), (len(counts), 1)), None

        def tree_set_targets_batch(self, targets):
            self.calls.append(("targets", [len(t) for t in targets]))

        def tree_register_batch(self, rot, t, scale, lambda_c, maxiter, tol):
            B = len(rot)
            # the "registration" marks every pair with the length of ... nothing: identity, q = pair count so far
            return rot, t, np.full(B, 3, np.int32), np.arange(B, dtype=float), np.zeros(B, np.int32), None

    rs = np.random.RandomState(0)
    sizes = [700, 710, 720, 730, 740]
    kinds = [np.float32, np.float64, np.float32, np.float64, np.float32]
    pairs = [(rs.rand(n, 3).astype(k), rs.rand(n + 5, 3)) for n, k in zip(sizes, kinds)]
    ctx = Recorder()
    res, info = H.registration_gmmtree_batch(pairs, maxiter=3, tol=1e-4, ctx=ctx, tree_level=2, return_info=True)
    assert len(res) == 5 and all(r is not None for r in res)
    builds = [c for c in ctx.calls if c[0] == "build"]
    assert sorted((b[1], tuple(b[2])) for b in builds) == [("float32", (700, 720, 740)), ("float64", (710, 730))]
    assert ctx.tree_dtype == np.dtype(np.float64)                         # restored
    assert info["build_iters"].shape == (5, 2) and info["registration_iters"] == [3] * 5 and info["status"] == [0] * 5
    # the q values number the pairs inside their own batch: float32 batch -> pairs 0, 2, 4 get 0, 1, 2
    assert [float(np.ravel(r.q)[0]) for r in res] == [0.0, 0.0, 1.0, 1.0, 2.0]
    # one kind only: one batch, no regrouping; an explicit pdf_dtype overrides the sources' types
    ctx2 = Recorder()
    H.registration_gmmtree_batch(pairs, maxiter=3, ctx=ctx2, tree_level=2, pdf_dtype=np.float64)
    assert [c for c in ctx2.calls if c[0] == "build"] == [("build", "float64", sizes)]
