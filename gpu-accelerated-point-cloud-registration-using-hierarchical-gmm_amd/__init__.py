"""hgmm_amd -- MI355X-native (gfx950) GMM / hierarchical-GMM EM engine.

Drop-in for the EM hot path of somanshu25/GPU-Accelerated-Point-Cloud-Registration-Using-
Hierarchical-GMM: hand-written HIP kernels behind a C ABI (include/hgmm.h), bound with ctypes.
NumPy on the host, no PyTorch, no CPU fallback.

    from hgmm_amd.gmm_waymo.gmm import GMM_GPU                 # src/python/gmm_waymo/src/gmm.py
    from hgmm_amd.gmm_waymo.gmm_impl import train_gmm, e_step  # .../gmm_impl.py
    from hgmm_amd.gmmreg_gpu.gmm import GMM_GPU                # src/python/gmmreg_gpu/gmm.py
    from hgmm_amd.hgmm.hgmm_gpu import buildGMMTree, GMMTree, registration_gmmtree

The reference's scripts import its modules by their bare names (``from gmm import GMM_GPU``,
``import cost_functions as cf`` ...): ``hgmm_amd.install_dropin("gmm_waymo" | "gmmreg_gpu" | "hgmm")``
registers this package's mirrors under those names, see INTEGRATION.md section 1.
"""
from ._native import Context, DeviceArray, DeviceScalar, HgmmError, default_context, set_default_context, use_context, load_library  # noqa: F401
from ._flat import DevicePoints, asarray  # noqa: F401

from ._dropin import install_dropin, uninstall_dropin  # noqa: F401

__version__ = "0.2.0"
