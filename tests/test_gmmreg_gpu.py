"""End-to-end L2 GMMReg registration through the drop-in API with the mixtures fitted on the GPU
engine, against the transform the reference produced on the same clouds (golden)."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_registration_gmmreg_matches_reference_run():
    import hgmm_amd
    from hgmm_amd.gmmreg_gpu.gmmreg import registration_gmmreg, RigidGMMReg
    g = load_golden("gmmreg_l2.npz")
    src, tgt = g["reg_source"], g["reg_target"]
    reg = RigidGMMReg(src, n_gmm_components=30)
    feats = []
    orig = reg._feature_gen.compute
    reg._feature_gen.compute = lambda d: feats.append(orig(d)) or feats[-1]
    trace = []
    reg.set_callbacks([lambda tf: trace.append(tf.rot.copy())])
    tfm = reg.registration(tgt)
    # the GPU-fitted mixtures equal the reference's NumPy-fitted ones (same KMeans init, 10 EM its)
    np.testing.assert_allclose(feats[0][0], g["reg_mu_target"], atol=2e-4)
    np.testing.assert_allclose(feats[1][0], g["reg_mu_source"], atol=2e-4)
    np.testing.assert_allclose(feats[1][1], g["reg_phi_source"], atol=2e-4)
    assert len(trace) >= 1
    # ... and so does the recovered transform (BFGS amplifies the 1e-4 mixture differences a little)
    np.testing.assert_allclose(tfm.rot, g["reg_rot"], atol=5e-3)
    np.testing.assert_allclose(tfm.t, g["reg_t"], atol=2e-3)
    tf2 = registration_gmmreg(src, tgt, n_gmm_components=30)
    np.testing.assert_allclose(tf2.rot, tfm.rot, atol=1e-9)


def test_device_gauss_transform_and_l2_cost_match_reference():
    """hgmm_gauss_transform (csrc/gmmreg_kernels.hip) against the reference's own outputs: the same
    golden vectors the host NumPy form is pinned to (tests/test_gmmreg_cpu.py), float64 on both sides."""
    import hgmm_amd
    from hgmm_amd.gmmreg_gpu import cost_functions as cf, transforms as tf
    g = load_golden("gmmreg_l2.npz")
    ctx = hgmm_amd.default_context()
    gt = tf.GaussTransform(g["mu_t"], 0.5, ctx=ctx)
    np.testing.assert_allclose(gt.compute(g["mu_s"], g["phi_t"]), g["gt_1d"], rtol=1e-13)
    np.testing.assert_allclose(gt.compute(g["mu_s"], g["phi_t"] * g["mu_t"].T), g["gt_2d"], rtol=1e-13)
    f, grad = cf.compute_l2_dist(g["mu_s"], g["phi_s"], g["mu_t"], g["phi_t"], float(g["sigma"]), ctx=ctx)
    np.testing.assert_allclose(f, g["l2_f"], rtol=1e-13)
    np.testing.assert_allclose(grad, g["l2_g"], rtol=1e-12, atol=1e-14)
    c = cf.RigidCostFunction(ctx=ctx)
    for th, f_ref, g_ref in zip(g["theta"], g["cost_f"], g["cost_g"]):
        f, grad = c(th, g["mu_s"], g["phi_s"], g["mu_t"], g["phi_t"], float(g["sigma"]))
        np.testing.assert_allclose(f, f_ref, rtol=1e-12)
        np.testing.assert_allclose(grad, g_ref, rtol=1e-10, atol=1e-12)
    # the BFGS solve on the reference's mixtures lands on the reference's transform
    from hgmm_amd.gmmreg_gpu import gmmreg
    reg = gmmreg.L2DistRegistration(g["reg_source"], None, cf.RigidCostFunction(ctx=ctx))
    res = reg.optimise(g["reg_mu_source"].astype(np.float64), g["reg_phi_source"] * 1e3,
                       g["reg_mu_target"].astype(np.float64), g["reg_phi_target"] * 1e3, reg._cost_fn.initial())
    tfm = reg._cost_fn.to_transformation(res.x)
    np.testing.assert_allclose(tfm.rot, g["reg_rot"], atol=1e-7)
    np.testing.assert_allclose(tfm.t, g["reg_t"], atol=1e-7)


@pytest.mark.parametrize("n_c,n_p,n_w", [(1, 1, 1), (5, 70, 3), (129, 64, 4), (800, 800, 4), (3000, 1500, 8)])
def test_device_gauss_transform_sizes(n_c, n_p, n_w):
    """Ragged sizes across the 64-point workgroup / 128-centre tile / centre-split boundaries
    against the host NumPy form."""
    import time
    import hgmm_amd
    from hgmm_amd.gmmreg_gpu import transforms as tf
    ctx = hgmm_amd.default_context()
    rs = np.random.RandomState(n_c + n_p)
    centres, pts = rs.rand(n_c, 3), rs.rand(n_p, 3)
    w = rs.randn(n_w, n_c) if n_w > 1 else rs.randn(n_c)
    h = 0.2
    host = tf.GaussTransform(centres, h).compute(pts, w)
    ctx.gauss_transform(centres, pts, w, h)
    t0 = time.perf_counter()
    dev = ctx.gauss_transform(centres, pts, w, h)
    dt = time.perf_counter() - t0
    assert dev.shape == host.shape
    np.testing.assert_allclose(dev, host, rtol=1e-12, atol=1e-13 * np.abs(host).max())
    print("gauss transform %d x %d x %d weights: %.0f us on the device" % (n_c, n_p, n_w, dt * 1e6))
    with pytest.raises(hgmm_amd.HgmmError):
        ctx.gauss_transform(centres, pts, np.ones((9, n_c)), h)
