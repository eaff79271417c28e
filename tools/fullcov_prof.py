#!/usr/bin/env python3
"""The flat full-covariance EM at C3 size (N = 1e6, J = 800, float64), alone -- for profiling:
    HGMM_FT_DEBUG=1 python tools/fullcov_prof.py 3          per-phase clock64 split of full_fused_kernel (stderr)
    rocprofv3 --pmc ... -- python tools/fullcov_prof.py 6   SQ counters of the kernel"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hgmm_amd  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
J = int(sys.argv[3]) if len(sys.argv) > 3 else 800
ctx = hgmm_amd.Context(0)
P = np.random.RandomState(0).rand(N, 3).astype(np.float32).astype(np.float64)
idx = np.random.RandomState(100).choice(N, J, replace=False)
ctx.set_points(P)
ctx.fullcov_fit(J, 1e-30, 1e-4, P[idx], 0.01, 2)
ctx.profile_reset()
ctx.profile_enable(True)
t0 = time.perf_counter()
ctx.fullcov_fit(J, 1e-30, 1e-4, P[idx], 0.01, iters)
dt = time.perf_counter() - t0
ctx.profile_enable(False)
ms, n = ctx.profile_get("full_fused")
print("fullcov N=%d J=%d: %.3f ms per iteration, full_fused kernel %.3f ms avg over %d launches" % (N, J, dt * 1e3 / iters, ms / max(n, 1), n))
ctx.close()
