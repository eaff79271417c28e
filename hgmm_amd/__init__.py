"""Importable alias for the package directory
``gpu-accelerated-point-cloud-registration-using-hierarchical-gmm_amd/`` (whose name, fixed by
the project layout, is not a valid Python identifier).  ``import hgmm_amd`` loads that
directory as the package ``hgmm_amd``."""
import importlib.util
import os
import sys

_REAL = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                     "gpu-accelerated-point-cloud-registration-using-hierarchical-gmm_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_REAL, "__init__.py"),
                                               submodule_search_locations=[_REAL])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
