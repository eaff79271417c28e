"""Module-level drop-in -- CPU part: the literal import lines of
the reference's callers resolve to this package once ``hgmm_amd.install_dropin(family)`` has run.
The lines are quoted from the reference (file:line in the comments); nothing here needs a GPU."""
import re
import sys
import os

import numpy as np
import pytest

from conftest import ROOT
import hgmm_amd


@pytest.fixture(autouse=True)
def clean_aliases():
    hgmm_amd.uninstall_dropin()
    yield
    hgmm_amd.uninstall_dropin()


def test_gmm_waymo_import_lines():
    hgmm_amd.install_dropin("gmm_waymo")
    ns = {}
    exec("from gmm import GMM_CPU, GMM_Sklearn, GMM_GPU\n"       # run_gmm_static.py:5, run_gmm_waymo_gpu.py:5
         "from gmm_impl import predict\n"                         # run_gmm_static.py:6
         "from gmm_impl import train_gmm, init_gmm_params, timer, predict\n", ns)   # gmm_waymo/src/gmm.py:9
    from hgmm_amd.gmm_waymo import gmm, gmm_impl
    assert ns["GMM_GPU"] is gmm.GMM_GPU and ns["GMM_CPU"] is gmm.GMM_CPU and ns["GMM_Sklearn"] is gmm.GMM_Sklearn
    assert ns["predict"] is gmm_impl.predict and ns["train_gmm"] is gmm_impl.train_gmm
    # constructor signatures of the reference (gmm.py:30,47,104,121)
    g = ns["GMM_GPU"](n_gmm_components=50, max_iter=50, cov_type='spherical')
    assert (g._n_gmm_components, g.max_iter, g.tol, g.cov_type) == (50, 50, 1e-4, 'spherical')
    s = ns["GMM_Sklearn"](n_gmm_components=7, max_iter=50, cov_type='spherical')
    assert (s._n_gmm_components, s.max_iter, s.cov_type) == (7, 50, 'spherical')


def test_gmmreg_import_lines():
    hgmm_amd.install_dropin("gmmreg_gpu")
    ns = {}
    exec("import gmm as ft\n"                  # gmmreg.py:8
         "import cost_functions as cf\n"       # gmmreg.py:9
         "import transforms as tf\n"           # cost_functions.py:5
         "import so\n"                         # cost_functions.py:6
         "from gmm_impl import train_gmm, init_gmm_params, timer, predict\n"   # gmmreg_gpu/gmm.py:9
         "import gmmreg\n", ns)
    from hgmm_amd.gmmreg_gpu import gmm, cost_functions, transforms, so, gmmreg
    assert ns["ft"] is gmm and ns["cf"] is cost_functions and ns["tf"] is transforms and ns["so"] is so
    assert ns["gmmreg"] is gmmreg
    for name in ("GMM_GPU", "GMM_CPU", "OneClassSVM", "Feature"):            # gmmreg.py:126,144 use ft.<name>
        assert hasattr(ns["ft"], name), name
    for name in ("registration_gmmreg", "registration_svr", "RigidGMMReg", "RigidSVR", "L2DistRegistration"):
        assert hasattr(gmmreg, name), name
    assert callable(ns["cf"].RigidCostFunction) and callable(ns["cf"].compute_l2_dist)
    assert callable(ns["so"].diff_rot_from_quaternion) and callable(ns["tf"].GaussTransform)


def test_hgmm_import_lines_and_family_rules():
    hgmm_amd.install_dropin("hgmm")
    import hgmm_gpu                                                          # the reference's hgmm/hgmm_gpu.py
    for name in ("buildGMMTree", "gmmTreeRegESTep", "GMMTree", "registration_gmmtree", "RigidTransformation",
                 "MstepResult"):
        assert hasattr(hgmm_gpu, name), name
    hgmm_amd.install_dropin("gmm_waymo")                                     # hgmm coexists with one gmm family
    with pytest.raises(ImportError):
        hgmm_amd.install_dropin("gmmreg_gpu")                                # gmm / gmm_impl mean something else there
    with pytest.raises(ValueError):
        hgmm_amd.install_dropin("icp")
    hgmm_amd.install_dropin("gmmreg_gpu", force=True)
    import gmm
    assert gmm.__name__.endswith("gmmreg_gpu.gmm")
    hgmm_amd.uninstall_dropin()
    assert "gmm" not in sys.modules and "hgmm_gpu" not in sys.modules


def test_sklearn_backed_features_run_on_the_host():
    """GMM_Sklearn (gmm.py:29-44) and OneClassSVM (gmm.py:151-177) are third-party estimators on the host:
    they need no GPU; the tuples they return have the reference's shape."""
    hgmm_amd.install_dropin("gmm_waymo")
    from gmm import GMM_Sklearn, OneClassSVM
    rs = np.random.RandomState(0)
    X = rs.rand(5, 3)[rs.randint(5, size=600)] + 0.02 * rs.randn(600, 3)
    f = GMM_Sklearn(n_gmm_components=5, max_iter=20, cov_type='spherical')
    f.init()
    means, weights, covs, none = f.compute(X)
    assert means.shape == (5, 3) and weights.shape == (5,) and covs.shape == (5,) and none is None
    assert f.predict(X).shape == (600,)
    o = OneClassSVM(3, 0.1, gamma=0.5, nu=0.1)
    o.init()
    sv, coef = o(X)                                                          # Feature.__call__ -> compute
    assert sv.shape[1] == 3 and coef.shape == (sv.shape[0],)
    g0 = o._gamma
    o.annealing()
    assert o._gamma == g0 * 10.0


def test_cpu_names_announce_the_engine():
    from hgmm_amd.gmm_waymo.gmm import GMM_CPU, EngineNotice
    g = GMM_CPU(n_gmm_components=4)
    with pytest.warns(EngineNotice):
        g.init()


def test_integration_section1_names_the_tested_block():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = [b for b in re.findall(r"```python\n(.*?)```", md, flags=re.S) if "install_dropin" in b]
    assert len(blocks) == 1
    assert "from gmm import GMM_CPU, GMM_Sklearn, GMM_GPU" in blocks[0]
