"""Module-level drop-in: make the reference's own import lines resolve to this package.

The reference is a set of script directories, each put on ``sys.path`` by running a script inside it;
its callers import sibling modules by bare name:

    src/python/gmm_waymo/src/run_gmm_static.py:5-6     from gmm import GMM_CPU, GMM_Sklearn, GMM_GPU
                                                        from gmm_impl import predict
    src/python/gmmreg_gpu/gmmreg.py:8-9                 import gmm as ft ; import cost_functions as cf
    src/python/gmmreg_gpu/cost_functions.py:5-6         import transforms as tf ; import so
    src/python/hgmm (demo tails, probreg-style users)   import hgmm_gpu

``install_dropin(family)`` registers the mirror modules of one such directory in ``sys.modules`` under
exactly those names (the same module objects as ``hgmm_amd.<family>.<name>``, not copies), so the
callers run unchanged on the MI355X engine.  ``gmm`` / ``gmm_impl`` exist in two of the reference's
directories with different semantics (flavour W vs G), hence the explicit family.
"""
from __future__ import annotations

import importlib
import sys

FAMILIES = {
    # reference directory            bare module name -> mirror inside this package
    "gmm_waymo": {"gmm_impl": "gmm_waymo.gmm_impl", "gmm": "gmm_waymo.gmm"},
    "gmmreg_gpu": {"gmm_impl": "gmmreg_gpu.gmm_impl", "gmm": "gmmreg_gpu.gmm", "so": "gmmreg_gpu.so",
                   "transforms": "gmmreg_gpu.transforms", "cost_functions": "gmmreg_gpu.cost_functions",
                   "gmmreg": "gmmreg_gpu.gmmreg"},
    "hgmm": {"hgmm_gpu": "hgmm.hgmm_gpu"},
}
_installed: dict[str, object] = {}


def install_dropin(family: str, force: bool = False):
    """Register the mirrors of one reference directory under the reference's bare module names.

    Returns the dict ``{bare name: module}``.  Refuses to shadow a *different* module already imported
    under one of the names (e.g. the reference's own ``gmm``) unless ``force`` is given."""
    if family not in FAMILIES:
        raise ValueError("unknown family %r (one of %s)" % (family, ", ".join(sorted(FAMILIES))))
    pkg = __name__.rsplit(".", 1)[0]
    out = {}
    for bare, target in FAMILIES[family].items():
        mod = importlib.import_module("%s.%s" % (pkg, target))
        have = sys.modules.get(bare)
        if have is not None and have is not mod and not force:
            raise ImportError("a different module named %r is already imported (%s); pass force=True to "
                              "replace it" % (bare, getattr(have, "__file__", "?")))
        sys.modules[bare] = mod
        _installed[bare] = mod
        out[bare] = mod
    return out


def uninstall_dropin():
    """Remove every alias this module registered (test hygiene)."""
    for bare, mod in list(_installed.items()):
        if sys.modules.get(bare) is mod:
            del sys.modules[bare]
        del _installed[bare]
