"""Runs the ctypes binding shown in INTEGRATION.md verbatim (section 2) against the built library
and checks it against the oracle -- the documented stub has to work as printed."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from oracle import flat_em

pytestmark = pytest.mark.gpu


def test_documented_ctypes_stub_works():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", md, flags=re.S)
    stub = [b for b in blocks if "hgmm_flat_train" in b and "def fit" in b]
    assert len(stub) == 1
    lib_path = os.path.join(ROOT, "gpu-accelerated-point-cloud-registration-using-hierarchical-gmm_amd",
                            "libhgmm_hip.so")
    code = stub[0].replace('C.CDLL("libhgmm_hip.so")', 'C.CDLL(%r)' % lib_path)
    ns = {}
    exec(code, ns)
    rs = np.random.RandomState(0)
    X = (rs.rand(6, 3)[rs.randint(6, size=5000)] + 0.03 * rs.randn(5000, 3)).astype(np.float32)
    mu0, w0, cov0 = flat_em.seeded_init(X, 24, 1)
    inv, mu, w, cov, lls = ns["fit"](X, mu0, cov0, w0, max_iter=6, tol=0.0)
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    o = flat_em.train(f64(X), 6, 0.0, f64(mu0), f64(cov0), f64(w0), "diag", "W")
    assert len(lls) == 6
    np.testing.assert_allclose(lls, o[4], atol=3e-5)
    np.testing.assert_allclose(mu, o[1], atol=1e-5)
    labels = ns["predict"](inv, mu, w, len(X))
    assert labels.dtype == np.int64
    assert (labels != flat_em.predict(f64(X), f64(inv), f64(mu), f64(w), "diag", "W")).mean() < 1e-3
    assert ns["lib"].hgmm_destroy(ns["ctx"]) == 0
