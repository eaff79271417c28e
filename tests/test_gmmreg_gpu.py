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
