"""INTEGRATION.md section 1 executed verbatim on the GPU: ``install_dropin`` + the reference's own
import lines + the fit / predict sequence of run_gmm_static.py:35-49 (viewer removed), labels checked
against the oracle's predict on the fitted parameters; then the gmmreg_gpu and hgmm families through
their bare module names."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT
import hgmm_amd
from oracle import flat_em

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def clean_aliases():
    hgmm_amd.uninstall_dropin()
    yield
    hgmm_amd.uninstall_dropin()


def _blobs(n=6000, k=12, seed=3):
    rs = np.random.RandomState(seed)
    return rs.rand(k, 3)[rs.randint(k, size=n)] + 0.03 * rs.randn(n, 3)


def test_integration_section1_block_verbatim(capsys):
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = [b for b in re.findall(r"```python\n(.*?)```", md, flags=re.S) if "install_dropin" in b]
    assert len(blocks) == 1
    source_np = _blobs()
    np.random.seed(11)                                   # init_gmm_params draws with the global RNG (gmm_impl.py:34)
    ns = {"source_np": source_np}
    exec(blocks[0], ns)
    gmm, labels = ns["gmm"], ns["gmm_idxs"]
    assert labels.shape == (len(source_np),) and labels.dtype == np.int64
    clf = gmm._clf
    assert clf.means_.shape == (50, 3) and clf.covariances_.shape == (50,) and len(clf.lls) >= 1
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    ref = flat_em.predict(f64(source_np.astype(np.float32)), f64(clf.inv_covs), f64(clf.means_), f64(clf.weights_),
                          "spherical", "W")
    # labels may differ only at genuine near-ties of the top-2 responsibilities (hard parts)
    bad = np.flatnonzero(labels != ref)
    if len(bad):
        _, lr, _, _ = flat_em.e_step_full(f64(source_np.astype(np.float32))[bad], f64(clf.inv_covs), f64(clf.means_),
                                          f64(clf.weights_), "spherical", "W")
        assert flat_em.near_tie_mask(lr, 1e-5).all(), len(bad)
    assert len(bad) < 1e-3 * len(labels)
    assert "GPU GMM TRAIN" in capsys.readouterr().out    # the reference's timer line (gmm.py:87)


def test_gmmreg_family_through_bare_names():
    hgmm_amd.install_dropin("gmmreg_gpu")
    import gmmreg
    import transforms as tf
    src = _blobs(3000, 10, 5)
    th = np.deg2rad(12.0)
    rot = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    tgt = src @ rot.T + np.array([0.05, -0.02, 0.03])
    before = np.linalg.norm(src - tgt, axis=1).mean()
    res = gmmreg.registration_gmmreg(src, tgt, n_gmm_components=40)
    assert isinstance(res, tf.RigidTransformation)
    after = np.linalg.norm(res.transform(src) - tgt, axis=1).mean()
    assert after < 0.25 * before, (before, after)
    # support-vector variant (gmmreg.py:159-169): same cost path, scikit-learn feature on the host
    res2 = gmmreg.registration_svr(src[::3], tgt[::3])
    after2 = np.linalg.norm(res2.transform(src) - tgt, axis=1).mean()
    assert np.isfinite(after2) and after2 < before, (before, after2)


def test_hgmm_family_through_bare_names():
    hgmm_amd.install_dropin("hgmm")
    import hgmm_gpu
    src = _blobs(4000, 9, 7)
    th = np.deg2rad(8.0)
    rot = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    tgt = src @ rot.T + np.array([0.01, 0.02, -0.01])
    res = hgmm_gpu.registration_gmmtree(src, tgt, maxiter=20, tol=1e-4, tree_level=2)
    moved = res.transformation.transform(src)
    assert np.linalg.norm(moved - tgt, axis=1).mean() < 0.3 * np.linalg.norm(src - tgt, axis=1).mean()
