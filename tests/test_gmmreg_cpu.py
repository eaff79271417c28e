"""Host-side L2 GMMReg path against vectors produced by the reference's own
cost_functions.py / transforms.py / so.py / gmmreg.py (tests/golden/gmmreg_l2.npz).  CPU only."""
import numpy as np

from conftest import load_golden


def test_quaternion_jacobian_matches_reference():
    from hgmm_amd.gmmreg_gpu import so
    g = load_golden("gmmreg_l2.npz")
    for q, d, r in zip(g["q"], g["d_rot"], g["rot"]):
        np.testing.assert_allclose(so.quaternion_matrix(q)[:3, :3], r, atol=1e-14)
        np.testing.assert_allclose(so.diff_rot_from_quaternion(q), d, rtol=1e-12, atol=1e-13)
    # the quirk-free form is the true Jacobian (finite differences), also for non-unit q
    q = np.array([0.8, 0.2, -0.4, 0.7])
    d = so.diff_rot_from_quaternion(q, reference_quirks=False)
    for k in range(4):
        e = np.zeros(4); e[k] = 1e-6
        fd = (so.quaternion_matrix(q + e)[:3, :3] - so.quaternion_matrix(q - e)[:3, :3]) / 2e-6
        np.testing.assert_allclose(d[k], fd, atol=1e-8)


def test_gauss_transform_and_l2_cost_match_reference():
    from hgmm_amd.gmmreg_gpu import cost_functions as cf, transforms as tf
    g = load_golden("gmmreg_l2.npz")
    gt = tf.GaussTransform(g["mu_t"], 0.5)
    np.testing.assert_allclose(gt.compute(g["mu_s"], g["phi_t"]), g["gt_1d"], rtol=1e-13)
    np.testing.assert_allclose(gt.compute(g["mu_s"], g["phi_t"] * g["mu_t"].T), g["gt_2d"], rtol=1e-13)
    f, grad = cf.compute_l2_dist(g["mu_s"], g["phi_s"], g["mu_t"], g["phi_t"], float(g["sigma"]))
    np.testing.assert_allclose(f, g["l2_f"], rtol=1e-13)
    np.testing.assert_allclose(grad, g["l2_g"], rtol=1e-12, atol=1e-14)
    c = cf.RigidCostFunction()
    assert np.array_equal(c.initial(), [1, 0, 0, 0, 0, 0, 0])
    for th, f_ref, g_ref in zip(g["theta"], g["cost_f"], g["cost_g"]):
        f, grad = c(th, g["mu_s"], g["phi_s"], g["mu_t"], g["phi_t"], float(g["sigma"]))
        np.testing.assert_allclose(f, f_ref, rtol=1e-12)
        np.testing.assert_allclose(grad, g_ref, rtol=1e-10, atol=1e-12)


def test_bfgs_on_reference_mixtures_reproduces_reference_transform():
    """Same mixtures in -> same rigid transform out as the reference's registration()."""
    from hgmm_amd.gmmreg_gpu import gmmreg, cost_functions as cf
    g = load_golden("gmmreg_l2.npz")
    reg = gmmreg.L2DistRegistration(g["reg_source"], None, cf.RigidCostFunction())
    np.testing.assert_allclose(reg._sigma, g["reg_sigma0"], rtol=1e-12)
    res = reg.optimise(g["reg_mu_source"].astype(np.float64), g["reg_phi_source"] * 1e3,
                       g["reg_mu_target"].astype(np.float64), g["reg_phi_target"] * 1e3,
                       reg._cost_fn.initial())
    tfm = reg._cost_fn.to_transformation(res.x)
    np.testing.assert_allclose(tfm.rot, g["reg_rot"], atol=1e-7)
    np.testing.assert_allclose(tfm.t, g["reg_t"], atol=1e-7)
